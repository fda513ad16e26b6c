"""Exhaustive scoring time at the SURVEY §8d shapes (all hypotheses live) — for A/B tests of score.hip builds."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pose_estimation_amd import ransac
dev = torch.device("cuda:0")
def run(name, f, work):
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e-3)
    t = float(np.median(ts[1:]))
    print(f"{name}: {t*1e6:9.1f} us  {work / t / 7.86432e13:.3f} of nominal VALU")
g = torch.Generator(device=dev).manual_seed(1)
B, H, N = 16, 4096, 8192
E = torch.randn((B, H, 9), device=dev, generator=g); pts = [torch.randn((B, N), device=dev, generator=g) * 0.3 for _ in range(4)]
run("E  B=16 H=4096 N=8192 ", lambda: ransac.score_essential(E, *pts, 1e-6), B * H * N * 20.0)
B, H, N = 32, 16384, 65536
P = torch.randn((B, H, 12), device=dev, generator=g); q = [torch.randn((B, N), device=dev, generator=g) for _ in range(5)]
run("PnP B=32 H=16384 N=65536", lambda: ransac.score_pnp(P, *q, 64.0), B * H * N * 17.0)
