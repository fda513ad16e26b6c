#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_pmc_wait.sh <tag> [pairs]
# Where the filter's wave cycles go: one SQ pass of the wave-state counters (quad-cycles: MI355X_MICROARCH.md, PMC slots)
# over the L2 matcher call at the bench's shape, summarised per kernel into gpurun_out/pmc_wait_<tag>_summary.json.
set -o pipefail
tag=$1; pairs=${2:-64}
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/pmcw_$tag
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS \
    --output-format csv -d /tmp/pmcw_$tag/p1 -o run -- python3 "$root/benchmarks/dbg/l2_kernel.py" $pairs > "$out/pmc_wait_${tag}.log" 2>&1 || { tail -5 "$out/pmc_wait_${tag}.log"; exit 1; }
cd "$root"
python3 tools/pmc_summary.py /tmp/pmcw_$tag l2_ > "$out/pmc_wait_${tag}_summary.json"
python3 - "$out/pmc_wait_${tag}_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k[:60], {c.replace("SQ_", ""): round(x["mean"]) for c, x in v.items()})
PY
