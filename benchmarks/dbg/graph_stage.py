import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pose_estimation_amd import matching, pipeline, ransac, synthetic
dev = torch.device("cuda:0")
_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
n, S, seed = 2048, 96, 31
thr_sq = float(np.float32((1.0 / synthetic.FX) ** 2))
runs = []
for v in range(3):
    sc = synthetic.two_view_pair(n=n, seed=500 + v)
    dq, dt, kq, kt = (_t(sc[k][None]) for k in ("desc1", "desc2", "kp1", "kp2"))
    o = pipeline.relative_pose_batch(lambda: (*matching.match_l2_ratio(dq, dt, 0.75), "l2_keep"), kq, kt, None, None, None, thr_sq,
                                     n_samples=S, seed=seed, want_points=True)
    runs.append((o["E_hyp"].clone(), [p.clone() for p in o["pts"]], o["n_match"].clone(), o["best_idx"].clone(), o["best_count"].clone()))
torch.cuda.synchronize()
E = runs[0][0].clone(); pts = [p.clone() for p in runs[0][1]]; nm = runs[0][2].clone()
def chain():
    Ec, slot_map, nl = ransac.compact_models(E)
    cnt = ransac.score_essential(Ec, *pts, thr_sq, nm, bailout=True)
    bi_c, bc = ransac.select_best(cnt)
    bi = ransac.lookup_index(slot_map, bi_c)
    return Ec, slot_map, nl, cnt, bi_c, bc, bi
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
s.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = chain()
for k, (Ek, pk, nk, bik, bck) in enumerate(runs + runs[:1]):
    E.copy_(Ek); nm.copy_(nk)
    for a, b in zip(pts, pk): a.copy_(b)
    g.replay(); torch.cuda.synchronize()
    ref = chain(); torch.cuda.synchronize()
    print(k, "graph bi/bc", out[6].item(), out[5].item(), "eager", ref[6].item(), ref[5].item(), "pipeline", bik.item(), bck.item(),
          "Ec", torch.equal(out[0], ref[0]), "slot", torch.equal(out[1], ref[1]), "nl", out[2].item(), ref[2].item(),
          "cnt", torch.equal(out[3], ref[3]), "cnt max", out[3].max().item(), ref[3].max().item())
