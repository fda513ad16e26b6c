"""Hamming matcher at 8192 x 8192: call time against the number of pairs (24 pairs = one round of resident blocks)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pose_estimation_amd import matching
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(7)
n, V = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 40
tab = torch.randint(0, 256, (V, n, 32), dtype=torch.uint8, generator=g).to(dev)
allp = [(i, j) for i in range(V) for j in range(i + 1, V)]
for B in (8, 16, 24, 32, 48, 64, 72, 96, 120, 240, 480):
    pairs = allp[:B]
    vq = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev)
    vt = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
    ts = []
    for k in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        matching.match_hamming(tab, tab, vq, vt)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    t = float(np.median(ts[2:]))
    print(f"{B:4d} pairs {n}x{n}: {t*1e3:8.1f} us  {t*1e3/B:6.2f} us/pair  frac {2.0 * n * n * 256 * B / (t * 1e-3) / 10066.3e12:.3f}")
