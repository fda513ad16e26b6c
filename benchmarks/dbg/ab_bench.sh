#!/bin/bash
# A/B of library builds on the whole step (bench.py, several streams): bash benchmarks/dbg/ab_bench.sh "<lib or -> ..." [streams] [rounds]
for r in $(seq ${3:-2}); do for l in $1; do
  if [ $l = - ]; then unset POSEHOT_LIB; else export POSEHOT_LIB=$PWD/benchmarks/dbg/$l; fi
  timeout -k 10 200 python3 -u bench.py --steps 30 --warmup 5 --cpu-pairs 0 --sustained-s 0 --streams ${2:-3} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lib=$l streams=${2:-3} pairs/s', round(d['value']), 'filter_us', round(r['dominant_kernel_us'],1), 'contended', round(r['contended_kernel_us'],1))" || exit 1
done; done
