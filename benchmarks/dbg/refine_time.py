"""ph_refine_pose_lm_soa alone at cfg 5's shape (32 frames x 65536 points, 80 % masked in) and at the board's (64 x 176);
prints the time, the iteration counts and a checksum of the refined poses (bit-compare library builds with POSEHOT_LIB)."""
import os, sys, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pose_estimation_amd import ransac
dev = torch.device("cuda:0")
for B, N in ((32, 65536), (64, 176)):
    rng = np.random.default_rng(5)
    Xw = rng.uniform(-1, 1, (B, N, 3)); Xw[..., 2] += 4.0
    rv = rng.normal(0, 0.2, (B, 3)); tv = rng.normal(0, 0.1, (B, 3))
    K4 = np.array([800.0, 800.0, 640.0, 360.0])
    def rod(w):
        th = np.linalg.norm(w); k = w / th; Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    uv = np.empty((B, N, 2))
    for b in range(B):
        Xc = Xw[b] @ rod(rv[b]).T + tv[b]
        uv[b, :, 0] = K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2]; uv[b, :, 1] = K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]
    uv += rng.normal(0, 0.5, uv.shape)
    mask_bits = rng.random((B, N)) < 0.8
    words = np.packbits(mask_bits.reshape(B, -1, 64), axis=-1, bitorder="little").view(np.uint64).reshape(B, -1).astype(np.int64) if N % 64 == 0 else None
    pose0 = np.concatenate([rv + rng.normal(0, 0.02, rv.shape), tv + rng.normal(0, 0.02, tv.shape)], axis=1)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    X, Y, Z, u, v = f(Xw[..., 0]), f(Xw[..., 1]), f(Xw[..., 2]), f(uv[..., 0]), f(uv[..., 1])
    m = torch.from_numpy(words).to(dev) if words is not None else None
    p0 = torch.from_numpy(pose0).to(dev); k4 = torch.from_numpy(K4).to(dev)
    ref = None
    for blocks in ("one", "auto"):
        ts = []
        for k in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            pose, rms, iters = ransac.refine_pose_lm_soa(X, Y, Z, u, v, p0, k4, mask=m, max_iters=30, blocks=blocks)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        h = hashlib.sha1(pose.cpu().numpy().tobytes() + rms.cpu().numpy().tobytes() + iters.cpu().numpy().tobytes()).hexdigest()[:12]
        it = iters.cpu().numpy()
        d = "" if ref is None else f"  max |pose - one| {float((pose - ref[0]).abs().max()):.2e}  iters equal {bool((iters == ref[1]).all())}"
        if ref is not None and not bool((iters == ref[1]).all()):
            print("   iters one :", ref[1].cpu().numpy().tolist()); print("   iters auto:", it.tolist())
        if ref is None:
            ref = (pose, iters)
        print(f"B={B} N={N} blocks={blocks}: {np.mean(ts[1:]):8.1f} us  iters {it.min()}..{it.max()} (mean {it.mean():.1f})  rms {float(rms.mean()):.4f}  sha1 {h}{d}")
