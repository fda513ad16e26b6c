#!/usr/bin/env python3
"""Launch the L2 matcher a few times at one shape (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose_estimation_amd import matching
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
tab = torch.randn((2 * P, 8192, 128), device=dev).abs_()
tab /= tab.norm(dim=2, keepdim=True)
vq = torch.arange(0, 2 * P, 2, dtype=torch.int32, device=dev); vt = vq + 1
nrm = matching.row_sqnorm(tab)
for _ in range(5):
    matching.match_l2(tab, tab, vq, vt, norm_q=nrm, norm_t=nrm)
torch.cuda.synchronize()
