"""ph_row_sqnorm at the bench's shape (64 x 8192 rows of 128 floats = 268 MB per call)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pose_estimation_amd import matching
dev = torch.device("cuda:0")
a = torch.rand((64, 8192, 128), device=dev)
ts = []
for k in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); n = matching.row_sqnorm(a); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
t = float(np.median(ts[2:]))
ref = (a[3, :100].double() ** 2).sum(1)
print(f"row_sqnorm {t:.1f} us, {a.numel() * 4 / t / 1e6:.2f} TB/s, max rel err vs f64 {float(((n[3, :100].double() - ref).abs() / ref).max()):.2e}")
