"""Hamming matcher time at cfg 3 / cfg 4 shapes (whole call: expand + matcher)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pose_estimation_amd import matching
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(7)
only = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # 32768 or 8192: that shape alone (counter passes)
for n, B in ((32768, 28), (8192, 496)):
    if only and n != only:
        continue
    V = 8 if n == 32768 else 32
    tab = torch.randint(0, 256, (V, n, 32), dtype=torch.uint8, generator=g).to(dev)
    pairs = [(i, j) for i in range(V) for j in range(i + 1, V)][:B]
    vq = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev)
    vt = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
    ts = []
    for k in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        matching.match_hamming(tab, tab, vq, vt)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    t = float(np.mean(ts[1:]))
    print(f"{len(pairs)} pairs {n}x{n}: {t:.3f} ms  {2.0 * n * n * 256 * len(pairs) / (t * 1e-3) / 1e15:.2f} POP/s")
