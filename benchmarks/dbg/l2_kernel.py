"""Filter-kernel time alone (events recorded by the library around l2_filter_kernel) for several batch sizes.
L2_MODE=top2 (default): ph_match_l2_top2 with ph_row_sqnorm's norms handed in; L2_MODE=ratio: ph_match_l2_ratio with the norms
computed inside the call (the pipelines' path: what tools/gpu_pmc.sh profiles for the bench line's traffic figure)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pose_estimation_amd import matching, synthetic
dev = torch.device("cuda:0")
hev = bench.HipEvents()
Ps = [int(a) for a in sys.argv[1:]] or [16, 32]
Pmax = max(Ps)
scs = [synthetic.two_view_pair(n=8192, seed=1002 + p) for p in range(min(Pmax, 8))]
d1 = torch.from_numpy(np.stack([scs[p % len(scs)]["desc1"] for p in range(Pmax)])).to(dev)
d2 = torch.from_numpy(np.stack([scs[p % len(scs)]["desc2"] for p in range(Pmax)])).to(dev)
for P in Ps:
    a, b = d1[:P].contiguous(), d2[:P].contiguous()
    nq, nt = matching.row_sqnorm(a), matching.row_sqnorm(b)
    ts = []
    for k in range(6):
        pr = hev.pair()
        torch.cuda.synchronize()
        if os.environ.get("L2_MODE", "top2") == "ratio":
            matching.match_l2_ratio(a, b, 0.75, kernel_events=pr)
        else:
            matching.match_l2(a, b, norm_q=nq, norm_t=nt, kernel_events=pr)
        torch.cuda.synchronize()
        ts.append(hev.elapsed_ms(pr) * 1e3)
    t = float(np.mean(ts[1:]))
    print(f"P={P:3d} filter {t:8.1f} us  per pair {t / P:6.2f} us  frac {2*8192*8192*128*P/(t*1e-6)/1e12/2516.6:.4f}")
