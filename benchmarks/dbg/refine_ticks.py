"""Where an LM iteration of refine_relative_pose_kernel goes: needs a build whose kernel reports thread 0's wall-clock ticks
(100 MHz) in the pass and in the serial part through the rms / n_inliers outputs (POSEHOT_LIB=benchmarks/dbg/librefdbg.so)."""
import numpy as np, torch
from pose_estimation_amd import ransac, synthetic

dev = torch.device("cuda", 0)
B, N = 64, 8192
sc = synthetic.two_view_pair(n=N, seed=5)
ok = sc["partner"] >= 0
m = int(ok.sum())
x1 = np.zeros((B, N), np.float32); y1 = x1.copy(); x2 = x1.copy(); y2 = x1.copy()
x1[:, :m] = sc["kp1"][ok, 0]; y1[:, :m] = sc["kp1"][ok, 1]
x2[:, :m] = sc["kp2"][sc["partner"][ok], 0]; y2[:, :m] = sc["kp2"][sc["partner"][ok], 1]
nv = np.full(B, m, np.int32)
R = sc["R"] @ synthetic.rot_y(0.002); t = sc["t"] / np.linalg.norm(sc["t"]) + [0.002, -0.001, 0.001]
tab = np.zeros((B, 16), np.float32); tab[:, :9] = R.reshape(9); tab[:, 9:12] = t
d = [torch.from_numpy(a).to(dev) for a in (x1, y1, x2, y2)]
thr = (3.0 / synthetic.FX) ** 2
for _ in range(3):
    _, rms, nin, iters = ransac.refine_relative_pose(*d, torch.from_numpy(tab).to(dev), thr, torch.from_numpy(nv).to(dev),
                                                    max_iters=4, want_stats=True)
torch.cuda.synchronize()
v = rms.cpu().numpy(); w = nin.cpu().numpy().astype(np.int64)
print("loop ticks per wave (sum over passes), block 0..3:")
for b in range(4):
    print(" w0", int(v[b] % 1e4), "w1", int(v[b] // 1e4 % 1e4), "w4", int(v[b] // 1e8 % 1e4), "w7", int(v[b] // 1e12), "w2", int(w[b] % 10000), "w5", int(w[b] // 10000))
