#!/bin/bash
# A/B timing of library builds on one box: bash benchmarks/dbg/ab.sh "<lib or -> ..." [rounds]; "-" = the in-tree library
for r in $(seq ${2:-2}); do for l in $1; do
  if [ $l = - ]; then unset POSEHOT_LIB; else export POSEHOT_LIB=$PWD/benchmarks/dbg/$l; fi
  echo "lib=$l $(timeout -k 10 100 python3 benchmarks/dbg/l2_kernel.py 64 | tail -1)" || exit 1
done; done
