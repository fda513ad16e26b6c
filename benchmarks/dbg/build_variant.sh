#!/bin/bash
# Build a variant of the library for A/B timing (runs here, on the CPU box; the .so travels with gpurun):
#   bash benchmarks/dbg/build_variant.sh <name> <file.hip> [extra hipcc flags...]
# e.g.  bash benchmarks/dbg/build_variant.sh ab_1 match_l2_filter.hip -DPH_ABLATE=1     -> benchmarks/dbg/libab_1.so
# The variant replaces ONE object of the in-tree build (run `make` in pose_estimation_amd/csrc first) and is selected
# at run time with POSEHOT_LIB=<path> (pose_estimation_amd/_lib.py); ablate.sh / ab.sh / ablate_pmc.sh expect these names.
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
cs=$root/pose_estimation_amd/csrc
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
case $src in match_l2_filter.hip|match_hamming.hip) flags="$flags -fno-honor-nans";; fivepoint.hip|refine_e.hip) flags="$flags -ffp-contract=fast";; score.hip) flags="$flags -fno-slp-vectorize";; esac
obj=/tmp/variant_$name.o
/opt/rocm/bin/hipcc $flags "$@" -c "$cs/$src" -o $obj
objs=""
for o in api match_hamming match_l2 match_l2_filter score triangulate board fivepoint p3p refine_e; do
    if [ "$o.hip" = "$src" ]; then objs="$objs $obj"; else objs="$objs $cs/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$root/benchmarks/dbg/lib$name.so" $objs
echo "$root/benchmarks/dbg/lib$name.so"
