import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pose_estimation_amd import matching, pipeline, plan, synthetic
dev = torch.device("cuda:0")
_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
n, S, seed, V, B = 2048, 96, 31, 4, 1
scs = [synthetic.two_view_pair(n=n, seed=500 + v) for v in range(V)]
dq = _t(np.stack([s["desc1"] for s in scs])); dt = _t(np.stack([s["desc2"] for s in scs]))
kq = _t(np.stack([s["kp1"] for s in scs])); kt = _t(np.stack([s["kp2"] for s in scs]))
thr_sq = float(np.float32((1.0 / synthetic.FX) ** 2))
p = plan.RelativePosePlan(dq, dt, kq, kt, B, thr_sq, n_samples=S, seed=seed)
rng = np.random.default_rng(B)
for trip in range(3):
    v = rng.integers(0, V, B).astype(np.int32); key = rng.integers(0, 1 << 30, B).astype(np.int32)
    p.set_pairs(v, v, key)
    out = p.replay(); torch.cuda.synchronize()
    vq, k = _t(v), _t(key)
    match = lambda: (*matching.match_l2_ratio(dq, dt, 0.75, vq, vq), "l2_keep")
    want = pipeline.relative_pose_batch(match, kq, kt, vq, vq, None, thr_sq, n_samples=S, seed=seed, item_key=k, want_counts=True)
    torch.cuda.synchronize()
    print("trip", trip, v, key, p.view_q.cpu().numpy(), p.key.cpu().numpy())
    for name in ("n_match", "E_hyp", "n_solutions", "best_idx", "best_count", "mask", "front", "win", "table"):
        a, b = out[name], want[name]
        same = torch.equal(a, b) if a.dtype != torch.float32 else bool(np.array_equal(a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32)))
        print(trip, name, same, (a.float() - b.float()).abs().max().item() if not same else "")
    print(out["table"].cpu().numpy()[0, 9:], want["table"].cpu().numpy()[0, 9:])
