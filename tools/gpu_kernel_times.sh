export TMPDIR=/tmp; R=$PWD; cd /tmp && rm -rf /tmp/kt && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python3 $R/bench.py --streams 1 --steps 30 --cpu-pairs 0 --sustained-s 0 --no-other-configs > /dev/null 2>&1; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kt_kernel_stats.csv; python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:${1:-8}]: print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
"
