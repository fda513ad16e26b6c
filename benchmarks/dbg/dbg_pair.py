import numpy as np, torch, sys
sys.path.insert(0, ".")
from pose_estimation_amd import geometry, matching, pipeline, ransac, synthetic
from pose_estimation_amd.allpairs import AllPairs
dev = torch.device("cuda:0")
V, N_FEAT = 40, 8192
THR_SQ = float(np.float32((1.5 / synthetic.FX) ** 2))
sc = synthetic.float_views(V, N_FEAT, seed=1004, config=4)
desc = torch.from_numpy(sc["desc"]).to(dev); kp = torch.from_numpy(sc["kp"]).to(dev)
norms = matching.row_sqnorm(desc)
vi, vj = synthetic.all_pairs(V)
p = 314
q = torch.tensor([vi[p]], dtype=torch.int32, device=dev); t = torch.tensor([vj[p]], dtype=torch.int32, device=dev)
key = torch.tensor([p], dtype=torch.int32, device=dev)
match = lambda: (*matching.match_l2(desc, desc, q, t, norm_q=norms, norm_t=norms), "l2")
out = pipeline.relative_pose_batch(match, kp, kp, q, t, None, THR_SQ, ratio=0.8, n_samples=96, seed=AllPairs.SEED, item_key=key, refine=False, want_points=True)
R, tt = synthetic.relative_pose(sc["R"][vi[p]], sc["t"][vi[p]], sc["R"][vj[p]], sc["t"][vj[p]])
def err(tab): return geometry.rotation_angle_deg(tab[0,:9].reshape(3,3).cpu().numpy(), R), geometry.direction_angle_deg(tab[0,9:12].cpu().numpy(), tt)
print("start", err(out["table"]), out["table"][0,12:].tolist())
pts = out["pts"]; nm = out["n_match"]
for sched in ([(3,4)], [(3,20)], [(10,3),(3,4)], [(20,3),(6,3),(3,4)], [(30,4),(3,4)], [(100,4),(10,4),(3,4)]):
    tab = out["table"].clone()
    for trunc, iters in sched:
        _, rms, nin, it = ransac.refine_relative_pose(*pts, tab, trunc*trunc*THR_SQ, n_valid=nm, max_iters=iters, want_stats=True)
        print("  ", trunc, iters, err(tab), float(rms[0])*synthetic.FX, int(nin[0]), int(it[0]))
    print(sched, err(tab))
# truth check: how many matches are inliers to the true pose at 3 thr
tab = out["table"].clone()
tab[0,:9] = torch.from_numpy(R.reshape(9).astype(np.float32)).to(dev); tab[0,9:12] = torch.from_numpy((tt/np.linalg.norm(tt)).astype(np.float32)).to(dev)
for iters in (0, 10):
    t2 = tab.clone()
    _, rms, nin, it = ransac.refine_relative_pose(*pts, t2, 9*THR_SQ, n_valid=nm, max_iters=iters, want_stats=True)
    print("from truth", iters, err(t2), float(rms[0])*synthetic.FX, int(nin[0]), int(it[0]))
n = int(nm[0])
x1,y1,x2,y2 = (a[0,:n].cpu().numpy().astype(np.float64) for a in pts)
print("n", n, "x1 range", x1.min(), x1.max(), y1.min(), y1.max())
# depth of points? check ids: fraction of true matches
