"""Kernel times of one ph_match_l2_top2 call at the bench's shape (torch events around the whole call minus the library's own
filter-kernel events = prep + presplit + re-rank)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pose_estimation_amd import matching, synthetic
dev = torch.device("cuda:0")
hev = bench.HipEvents()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scs = [synthetic.two_view_pair(n=8192, seed=1002 + p) for p in range(8)]
a = torch.from_numpy(np.stack([scs[p % 8]["desc1"] for p in range(P)])).to(dev)
b = torch.from_numpy(np.stack([scs[p % 8]["desc2"] for p in range(P)])).to(dev)
nq, nt = matching.row_sqnorm(a), matching.row_sqnorm(b)
ts, tf = [], []
for k in range(6):
    pr = hev.pair()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    matching.match_l2(a, b, norm_q=nq, norm_t=nt, kernel_events=pr)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3); tf.append(hev.elapsed_ms(pr) * 1e3)
print(f"call {np.mean(ts[1:]):.1f} us, filter {np.mean(tf[1:]):.1f} us, rest (prep + presplit + rerank) {np.mean(ts[1:]) - np.mean(tf[1:]):.1f} us")
