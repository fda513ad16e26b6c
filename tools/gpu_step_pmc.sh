#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_step_pmc.sh <tag>
# SQ counters of EVERY kernel of a bench step (one stream, 64 pairs, 3 timed steps), summarised per kernel by
# tools/pmc_summary.py into gpurun_out/step_pmc_<tag>_summary.json.
set -o pipefail
tag=$1
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/step_pmc_$tag
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d /tmp/step_pmc_$tag -o run -- python3 "$root/bench.py" --streams 1 --steps 3 --warmup 1 --cpu-pairs 0 --sustained-s 0 --no-other-configs \
    > "$out/step_pmc_${tag}.log" 2>&1 || { tail -5 "$out/step_pmc_${tag}.log"; exit 1; }
cd "$root"
python3 tools/pmc_summary.py /tmp/step_pmc_$tag > "$out/step_pmc_${tag}_summary.json"
python3 - "$out/step_pmc_${tag}_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", {"mean": 0})["mean"]):
    print(k[:58].ljust(58), {c.replace("SQ_", ""): round(x["mean"]) for c, x in v.items() if c in ("SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")})
PY
