#!/usr/bin/env python3
"""The multi-view configurations of BASELINE.json through the relative-pose pipeline.

    python benchmarks/configs.py --cfg 3            # 100-camera ring, 4,950 pairs, 32k binary descriptors
    python benchmarks/configs.py --cfg 4 --pairs N  # 1,000 views, 8k binary descriptors, first N of 499,500 pairs

Prints one JSON line: pairs/s over the whole pair list (views resident, pairs in chunks, the front of chunk
k+1 enqueued before the host finishes chunk k) and the accuracy against the synthetic ground truth.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose_estimation_amd import geometry, matching, pipeline, sharding, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=3)
    ap.add_argument("--views", type=int, default=0)
    ap.add_argument("--feat", type=int, default=0)
    ap.add_argument("--pairs", type=int, default=0, help="limit the pair list (0 = all)")
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--samples", type=int, default=410)
    args = ap.parse_args()
    V = args.views or (100 if args.cfg == 3 else 1000)
    n = args.feat or (32768 if args.cfg == 3 else 8192)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)

    t0 = time.perf_counter()
    sc = synthetic.binary_views(V, n, seed=1000 + args.cfg, config=args.cfg)
    gen_s = time.perf_counter() - t0
    desc = torch.from_numpy(sc["desc"]).to(dev)
    kp = torch.from_numpy(sc["kp"]).to(dev)
    vi, vj = synthetic.all_pairs(V)
    if args.pairs:
        vi, vj = vi[:args.pairs], vj[:args.pairs]
    mine = sharding.shard_indices(len(vi), rank, world).numpy()         # block-cyclic shard of the pair list
    vi, vj = vi[mine], vj[mine]
    thr_sq = float(np.float32((1.5 / synthetic.FX) ** 2))

    def front(s, e):
        q = torch.from_numpy(vi[s:e]).to(dev); t = torch.from_numpy(vj[s:e]).to(dev)
        return pipeline.relative_pose_front(lambda: (*matching.match_hamming(desc, desc, q, t), "hamming"), kp, kp, q, t,
                                            None, thr_sq, ratio=0.8, n_samples=args.samples, seed=s)

    tables = []
    f = front(0, min(args.chunk, len(vi)))             # warm-up chunk (also first real chunk's shapes)
    pipeline.relative_pose_back(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = None
    for s in range(0, len(vi), args.chunk):
        f = front(s, min(s + args.chunk, len(vi)))
        if pending is not None:
            tables.append(pipeline.relative_pose_back(pending)["table"])
        pending = f
    tables.append(pipeline.relative_pose_back(pending)["table"])
    table = torch.cat(tables)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    tab = table.cpu().numpy()
    rerr, terr = [], []
    for p in range(len(vi)):
        R, t = synthetic.relative_pose(sc["R"][vi[p]], sc["t"][vi[p]], sc["R"][vj[p]], sc["t"][vj[p]])
        rerr.append(geometry.rotation_angle_deg(tab[p, :9].reshape(3, 3), R))
        terr.append(geometry.direction_angle_deg(tab[p, 9:12], t))
    rerr, terr = np.array(rerr), np.array(terr)
    print(json.dumps({"cfg": args.cfg, "views": V, "features": n, "pairs": int(len(vi)), "rank": rank, "world": world,
                      "seconds": dt, "pairs_per_s": len(vi) / dt, "hyps_per_s": len(vi) * 10 * args.samples / dt,
                      "scene_generation_s": gen_s, "median_matches": float(np.median(tab[:, 13])),
                      "median_inliers": float(np.median(tab[:, 12])),
                      "rot_err_deg": {"median": float(np.median(rerr)), "p95": float(np.percentile(rerr, 95)), "max": float(rerr.max())},
                      "t_dir_err_deg": {"median": float(np.median(terr)), "p95": float(np.percentile(terr, 95)), "max": float(terr.max())},
                      "ok_within_1deg": float(np.mean((rerr < 1.0) & (terr < 1.0)))}))


if __name__ == "__main__":
    main()
