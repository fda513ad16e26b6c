import numpy as np, torch, sys
sys.path.insert(0, ".")
from pose_estimation_amd import geometry, matching, pipeline, sharding, synthetic
from pose_estimation_amd.allpairs import AllPairs
dev = torch.device("cuda:0")
V, N_FEAT, CHUNK = 40, 8192, 64
THR_SQ = float(np.float32((1.5 / synthetic.FX) ** 2))
sc = synthetic.float_views(V, N_FEAT, seed=1004, config=4)
desc = torch.from_numpy(sc["desc"]).to(dev); kp = torch.from_numpy(sc["kp"]).to(dev)
norms = matching.row_sqnorm(desc)
vi, vj = synthetic.all_pairs(V)
for samples, iters, trunc in ((96, 4, 3.0), (96, 10, 3.0), (96, 20, 3.0), (96, 10, 5.0), (200, 4, 3.0)):
  pipeline.REFINE_ITERS = iters; pipeline.REFINE_TRUNC = trunc
  for refine in (True,):
    run = AllPairs(desc, kp, norms, THR_SQ, samples, True, refine=refine)
    full, _, _ = sharding.run_pair_list(vi, vj, run.chunk, 0, 1, chunk=CHUNK, sync=torch.cuda.synchronize, device=dev)
    tab = full.cpu().numpy()
    rerr = []; terr = []
    for p in range(len(vi)):
        R, t = synthetic.relative_pose(sc["R"][vi[p]], sc["t"][vi[p]], sc["R"][vj[p]], sc["t"][vj[p]])
        rerr.append(geometry.rotation_angle_deg(tab[p, :9].reshape(3, 3), R)); terr.append(geometry.direction_angle_deg(tab[p, 9:12], t))
    rerr = np.array(rerr); terr = np.array(terr)
    bad = np.where((rerr >= 1) | (terr >= 1))[0]
    print(samples, iters, trunc, "median", np.median(rerr), np.median(terr), "p95", np.percentile(rerr,95), np.percentile(terr,95), "max", rerr.max(), terr.max())
    for p in bad: print("  bad", p, vi[p], vj[p], rerr[p], terr[p], tab[p, 12:16])
