bash tools/gpu_bench_profile.sh r3i > gpurun_out/r3i_profile.log 2>&1
python3 - <<EOF
import json, csv
d=json.load(open("gpurun_out/bench_r3i.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["vendor_gemm_same_box"]["frac_of_peak"])
s=json.load(open("gpurun_out/bench_r3i_streams1.json")); print("streams1", s["value"], s["ms_per_step"], "vendor" in str(s["roofline"].keys()))
for r in list(csv.DictReader(open("gpurun_out/bench_r3i_streams1_kernel_stats.csv")))[:12]: print(r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
EOF
