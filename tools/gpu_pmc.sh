#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_pmc.sh <tag> [pairs]
# Counter passes over the L2 matcher call at the bench's shape (benchmarks/dbg/l2_kernel.py): one rocprofv3 run per
# counter group (the guide's limits: 8 SQ, FETCH_SIZE alone, WRITE_SIZE alone), summarised per kernel by
# tools/pmc_summary.py into gpurun_out/pmc_<tag>_summary.json.
set -o pipefail
tag=$1; pairs=${2:-64}
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/pmc_$tag
i=0
for ctrs in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$tag/p$i -o run -- python3 "$root/benchmarks/dbg/l2_kernel.py" $pairs > "$out/pmc_${tag}_p$i.log" 2>&1 || { tail -5 "$out/pmc_${tag}_p$i.log"; exit 1; }
done
cd "$root"
python3 tools/pmc_summary.py /tmp/pmc_$tag l2_ > "$out/pmc_${tag}_summary.json"
python3 - "$out/pmc_${tag}_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k[:60], {c: round(x["mean"]) for c, x in v.items()})
PY
