#!/usr/bin/env python3
"""Launch the L2 matcher a few times at one shape (for rocprofv3 passes) and print HIP-event times.

    python3 benchmarks/l2_once.py [pairs] [precision]     # precision 0 = f32 MFMA, 1 = fp16 filter + re-rank
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose_estimation_amd import matching
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
tab = torch.randn((2 * P, 8192, 128), device=dev).abs_()
tab /= tab.norm(dim=2, keepdim=True)
vq = torch.arange(0, 2 * P, 2, dtype=torch.int32, device=dev); vt = vq + 1
nrm = matching.row_sqnorm(tab)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
for k in range(6):
    ev[k].record()
    out = matching.match_l2(tab, tab, vq, vt, norm_q=nrm, norm_t=nrm, precision=prec)
ev[6].record()
torch.cuda.synchronize()
ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(6)]
print(f"pairs={P} precision={prec} ms per call: " + " ".join(f"{m:.3f}" for m in ms) +
      f"  -> {2 * 8192 * 8192 * 128 * P / (min(ms) * 1e-3) / 1e12:.1f} algorithmic TFLOP/s")
