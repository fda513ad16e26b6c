#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_cfg_traces.sh <tag>
# rocprofv3 kernel statistics of configs 3, 4 (binary and f32) and 5 on one stream, copied to gpurun_out/cfg{3,4b,4f,5}_<tag>_kernel_stats.csv
# (the per-kernel averages the numbers of DESIGN 3.4 are made of).
tag=$1; export TMPDIR=/tmp; R=$PWD
run() {   # name, args...
    local name=$1; shift
    cd /tmp && rm -rf /tmp/ct_$name && timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ct_$name -o run -- python3 $R/benchmarks/configs.py "$@" > $R/gpurun_out/${name}_${tag}_traced.jsonl 2> $R/gpurun_out/${name}_${tag}.err || return 1
    local f=$(find /tmp/ct_$name -name "*kernel_stats.csv" | head -1)
    cp "$f" $R/gpurun_out/${name}_${tag}_kernel_stats.csv
    echo "== $name"; python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:8]: print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
"
}
run cfg3 --cfg 3 --samples 920 --streams 1 && run cfg4b --cfg 4 --pairs 20000 --samples 920 --streams 1 && run cfg4f --cfg 4 --pairs 20000 --samples 920 --streams 1 --f32 && run cfg5 --cfg 5
