# per-launch durations of the kernels whose name contains $1 (default score_kernel) in one-stream bench steps
export TMPDIR=/tmp; R=$PWD; cd /tmp && rm -rf /tmp/ktr && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr -o run -- python3 $R/bench.py --streams 1 --steps 10 --cpu-pairs 0 --sustained-s 0 --no-other-configs > /dev/null 2>&1; f=$(find /tmp/ktr -name "*kernel_trace.csv" | head -1); python3 -c "
import csv
rows=[r for r in csv.DictReader(open('$f')) if '${1:-score_kernel}' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-12:]: print(r['Kernel_Name'][:60], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
"
