# kernel statistics of single-pair batches (benchmarks/latency.py --pairs 1): which kernels make up one pair's latency
export TMPDIR=/tmp; R=$PWD; cd /tmp && rm -rf /tmp/lat && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lat -o run -- python3 $R/benchmarks/latency.py --pairs 1 --iters 50 > $R/gpurun_out/latency_b1.json 2>/dev/null; f=$(find /tmp/lat -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/latency_b1_kernel_stats.csv; python3 -c "
import csv
tot=0
for r in list(csv.DictReader(open('$f')))[:${1:-22}]: print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
"; cat $R/gpurun_out/latency_b1.json | cut -c1-400
