#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/gpu_bench_profile.sh <tag>
# Writes gpurun_out/bench_<tag>.json (3 streams, the headline), bench_<tag>_streams1.json and the rocprofv3 kernel
# statistics of the same commands: bench_<tag>_kernel_stats.csv, bench_<tag>_streams1_kernel_stats.csv.
set -o pipefail
tag=$1
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 280 python3 -u bench.py > "$out/bench_${tag}.json" 2> "$out/bench_${tag}.err" || exit 1
timeout -k 10 200 python3 -u bench.py --streams 1 --cpu-pairs 0 --no-other-configs > "$out/bench_${tag}_streams1.json" 2>> "$out/bench_${tag}.err" || exit 1
cd /tmp || exit 1
for s in 1 3; do
    d=/tmp/prof_${tag}_$s
    rm -rf "$d"
    timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o run -- python3 "$root/bench.py" --streams $s --cpu-pairs 0 --sustained-s 0 --no-other-configs \
        > "$out/bench_${tag}_profiled_s$s.json" 2>> "$out/bench_${tag}.err" || exit 1
    f=$(find "$d" -name "*kernel_stats.csv" | head -1)
    if [ "$s" = 1 ]; then cp "$f" "$out/bench_${tag}_streams1_kernel_stats.csv"; else cp "$f" "$out/bench_${tag}_kernel_stats.csv"; fi
done
cd "$root"
head -c 400 "$out/bench_${tag}.json"; echo
head -25 "$out/bench_${tag}_streams1_kernel_stats.csv"
