# kernel statistics of cfg 5 (32 frames x 65 536 correspondences, P3P RANSAC + LM): what besides the scoring kernel is in a batch
export TMPDIR=/tmp; R=$PWD; cd /tmp && rm -rf /tmp/c5 && timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5 -o run -- python3 $R/benchmarks/configs.py --cfg 5 > $R/gpurun_out/cfg5_traced.jsonl 2>/dev/null; f=$(find /tmp/c5 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/cfg5_kernel_stats.csv; python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:${1:-14}]: print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
"
