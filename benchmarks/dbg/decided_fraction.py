"""How many queries of the bench's pairs can ph_match_l2_ratio decide from approximate scores?  (float64 scores as a stand-in
for the filter's approximate ones; e from the kernel's formula.)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pose_estimation_amd import synthetic
dev = torch.device("cuda:0")
sc = synthetic.two_view_pair(n=8192, seed=1002)
q = torch.from_numpy(sc["desc1"]).to(dev).double(); t = torch.from_numpy(sc["desc2"]).to(dev).double()
qn = (q * q).sum(1); tn = (t * t).sum(1)
s = q @ t.T - 0.5 * tn[None]
top = torch.topk(s, 3, dim=1).values
Tmax, Qmax = tn.max().sqrt(), qn.max().sqrt()
e = (2.0 ** -10 + 2.0 ** -13) * (qn.sqrt() * Tmax + 0.5 * Tmax ** 2) + 2.0 ** -23 * Tmax * (qn.sqrt() + Qmax)
a1, a2 = top[:, 0], top[:, 1]
sep = (a1 - a2) > 2 * e
d1h = (qn - 2 * a1 + 2 * e).clamp(min=0); d1l = (qn - 2 * a1 - 2 * e).clamp(min=0)
d2h = (qn - 2 * a2 + 2 * e).clamp(min=0); d2l = (qn - 2 * a2 - 2 * e).clamp(min=0)
r2 = 0.75 ** 2
keep_sure = d1h < r2 * d2l; drop_sure = d1l > r2 * d2h
print(f"e mean {e.mean().item():.2e}; top separated {sep.float().mean().item():.3f}; keep sure {keep_sure.float().mean().item():.3f}; "
      f"drop sure {drop_sure.float().mean().item():.3f}; decided {(sep & (keep_sure | drop_sure)).float().mean().item():.3f}; "
      f"gap a1-a2 median {(a1 - a2).median().item():.2e}, a2-a3 median {(top[:, 1] - top[:, 2]).median().item():.2e}")
