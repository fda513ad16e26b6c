# kernel statistics of a cfg 4 run (binary unless $2 = --f32): what a chunk of 64 pairs is made of
export TMPDIR=/tmp; R=$PWD; cd /tmp && rm -rf /tmp/c4 && timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4 -o run -- python3 $R/benchmarks/configs.py --cfg 4 --pairs 20000 --samples 920 --streams 1 $2 > $R/gpurun_out/cfg4_traced.jsonl 2>/dev/null; f=$(find /tmp/c4 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/cfg4_kernel_stats.csv; python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:${1:-14}]: print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
"
