#!/usr/bin/env python3
"""Candidate-list statistics of the fp16 candidate filter (reads the workspace the C ABI call leaves behind)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose_estimation_amd import _lib as L, matching, synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = sys.argv[2] if len(sys.argv) > 2 else "random"
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
n = 8192
if kind == "random":
    tab = torch.randn((2 * P, n, 128), device=dev).abs_()
    tab /= tab.norm(dim=2, keepdim=True)
else:
    rng = np.random.default_rng(0)
    views = []
    for p in range(P):
        sc = synthetic.two_view_pair(n=n, seed=100 + p)
        views += [sc["desc1"], sc["desc2"]]
    tab = torch.from_numpy(np.stack(views)).to(dev)
vq = torch.arange(0, 2 * P, 2, dtype=torch.int32, device=dev); vt = vq + 1
nrm = matching.row_sqnorm(tab)
lib = L.load()
wsb = lib.ph_match_l2_ws_bytes(P, n, n, 128, prec)
ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
idx = torch.empty((P, n), dtype=torch.int32, device=dev); d1 = torch.empty((P, n), device=dev); d2 = torch.empty_like(d1)
L.check(lib.ph_match_l2_top2(L.ptr(tab), L.ptr(tab), L.ptr(nrm), L.ptr(nrm), L.ptr(vq), L.ptr(vt), None, None, P, n, n, 128, prec,
                             L.ptr(idx), L.ptr(d1), L.ptr(d2), L.ptr(ws), wsb, L.stream_ptr()), "l2")
torch.cuda.synchronize()
off = (2 * P * 4 + 255) // 256 * 256
slots = 2 * min(max(1, -(-512 // (P * (n // 256)))), -(-n // 1024))   # plan_l2_filter: two 4-wave blocks per CU, >= 16 staged tiles per split
cnt = ws[off:off + P * n * slots * 4].view(torch.int32).cpu().numpy().reshape(P * n, slots)
print(f"slots {slots}: per-segment max {cnt.max()} (capacity 32), overflowing queries {(cnt.max(1) > 32).mean():.5f}")
cnt = cnt.sum(1)
print(f"{kind}: candidates per query mean {cnt.mean():.2f} median {np.median(cnt):.0f} p99 {np.percentile(cnt, 99):.0f} max {cnt.max()}")
