#!/bin/bash
# timing-only ablations of l2_filter_kernel (results are wrong by construction; build the variants first:
#   for a in 1 5 15; do bash benchmarks/dbg/build_variant.sh ab_$a match_l2_filter.hip -DPH_ABLATE=$a; done): bit 1 = no score processing, 2 = no LDS
# fragment reads, 4 = no LDS-DMA staging, 8 = no barrier
for a in ${ABL:-0 1 5}; do
  if [ $a = 0 ]; then unset POSEHOT_LIB; else export POSEHOT_LIB=$PWD/benchmarks/dbg/libab_$a.so; fi
  echo "ablate=$a"; timeout -k 10 100 python3 benchmarks/dbg/l2_kernel.py 64 || exit 1
done
