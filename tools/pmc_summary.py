#!/usr/bin/env python3
"""Mean per-launch value of every counter in rocprofv3 --pmc CSV outputs, per kernel.

    python3 tools/pmc_summary.py gpurun_out/pmc_dir [kernel-substring]
"""
import collections, csv, glob, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
out = {k: {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in d.items()} for k, d in acc.items() if sub in k}
print(json.dumps(out, indent=1))
