#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_pmc_hamming.sh <tag>
# Counter pass over the Hamming matcher at cfg 3's and cfg 4's shapes (benchmarks/dbg/ham_time.py), summarised per kernel
# by tools/pmc_summary.py into gpurun_out/pmc_ham_<tag>_summary.json (the launches of both shapes are averaged per kernel
# name; ham_time.py prints each shape's time).
set -o pipefail
tag=$1
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/pmc_ham_$tag
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_ham_$tag/p1 -o run -- python3 "$root/benchmarks/dbg/ham_time.py" 32768 > "$out/pmc_ham_${tag}_p1.log" 2>&1 || { tail -5 "$out/pmc_ham_${tag}_p1.log"; exit 1; }
cd "$root"
python3 tools/pmc_summary.py /tmp/pmc_ham_$tag hamming > "$out/pmc_ham_${tag}_summary.json"
python3 - "$out/pmc_ham_${tag}_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k[:60], {c: round(x["mean"]) for c, x in v.items()}, "launches", next(iter(v.values()))["launches"])
PY
