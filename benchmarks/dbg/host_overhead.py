"""Host enqueue cost of one relative-pose step: the same call chain on a tiny problem (GPU time ~ 0)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pose_estimation_amd import matching, pipeline, synthetic
dev = torch.device("cuda:0")
P, n = 2, 256
scs = [synthetic.two_view_pair(n=n, seed=p) for p in range(P)]
t = lambda k: torch.from_numpy(np.stack([s[k] for s in scs])).to(dev)
dq, dt, kq, kt = t("desc1"), t("desc2"), t("kp1"), t("kp2")
thr_sq = float((1.0 / synthetic.FX) ** 2)
def step(k):
    def match():
        nq = matching.row_sqnorm(dq); nt = matching.row_sqnorm(dt)
        return (*matching.match_l2(dq, dt, norm_q=nq, norm_t=nt), "l2")
    return pipeline.relative_pose_batch(match, kq, kt, None, None, None, thr_sq, n_samples=64, seed=k, want_counts=True)
for k in range(5): step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(200): step(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue per step: {(t1-t0)/200*1e3:.3f} ms; with final sync {(t2-t0)/200*1e3:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(100): step(k)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
