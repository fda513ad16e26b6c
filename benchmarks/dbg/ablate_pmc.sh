#!/bin/bash
# SQ counters of the ablated filter variants (see ablate.sh); usage: bash benchmarks/dbg/ablate_pmc.sh "15 5 1 0"
set -o pipefail
root=$(pwd); export TMPDIR=/tmp
for a in $1; do
  if [ $a = 0 ]; then unset POSEHOT_LIB; else export POSEHOT_LIB=$root/benchmarks/dbg/libab_$a.so; fi
  rm -rf /tmp/pmc_ab$a; cd /tmp
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_ab$a -o run -- python3 "$root/benchmarks/dbg/l2_kernel.py" 64 > "$root/gpurun_out/ab_pmc_$a.log" 2>&1 || { tail -5 "$root/gpurun_out/ab_pmc_$a.log"; exit 1; }
  cd "$root"
  echo "ablate=$a $(grep 'P= 64' gpurun_out/ab_pmc_$a.log)"
  python3 tools/pmc_summary.py /tmp/pmc_ab$a l2_filter_kernel | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print({c: round(x['mean']) for c,x in v.items()})"
done
