#!/usr/bin/env python3
"""End-to-end latency of small batches (BASELINE.json configs[1] read literally: ONE pair).

    python benchmarks/latency.py [--pairs 1 8] [--iters 50]

Per batch size: the eager pipeline (about thirty launches through Python wrappers that allocate their outputs) against
one replay of a RelativePosePlan (the same chain captured as a hipGraph, pose_estimation_amd/plan.py), each as
  latency   — submit, wait for the result, repeat (median over --iters), and
  pipelined — --iters batches back to back, one wait at the end.
One JSON line per batch size.  The workload per pair is bench.py's: 8192 x 128 f32 descriptors per view, 920 minimal
samples (>= 4096 real hypotheses), row norms computed inside the batch, refinement on."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose_estimation_amd import matching, pipeline, plan, synthetic  # noqa: E402


def measure(B, iters, dev, n=8192, S=920):
    scs = [synthetic.two_view_pair(n=n, seed=1002 + p) for p in range(B)]
    t = lambda k: torch.from_numpy(np.ascontiguousarray(np.stack([s[k] for s in scs]))).to(dev)
    dq, dt, kq, kt = t("desc1"), t("desc2"), t("kp1"), t("kp2")
    thr_sq = float(np.float32((1.0 / synthetic.FX) ** 2))
    vq = torch.arange(B, dtype=torch.int32, device=dev)

    def eager():
        match = lambda: (*matching.match_l2_ratio(dq, dt, 0.75, vq, vq), "l2_keep")       # row norms inside the call
        return pipeline.relative_pose_batch(match, kq, kt, vq, vq, None, thr_sq, n_samples=S, seed=1000, item_key=vq,
                                            want_counts=True)

    p = plan.RelativePosePlan(dq, dt, kq, kt, B, thr_sq, n_samples=S, seed=1000)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lat = []
        for _ in range(iters):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return float(np.median(lat)) * 1e6, (time.perf_counter() - t0) / iters * 1e6

    e_lat, e_pipe = timed(eager)
    g_lat, g_pipe = timed(p.replay)
    # the same replay for views that stay resident: their row norms computed once, when the plan is made (ph_row_sqnorm),
    # handed to the matcher — what a front end that matches the same views again would run
    pc = plan.RelativePosePlan(dq, dt, kq, kt, B, thr_sq, n_samples=S, seed=1000, fresh_norms=False)
    c_lat, c_pipe = timed(pc.replay)
    same = bool(np.array_equal(eager()["table"].cpu().numpy().view(np.uint32), p.replay()["table"].cpu().numpy().view(np.uint32)))
    real = float(p.out["n_solutions"].sum(dim=1).float().mean().item())
    return {"pairs_per_batch": B, "features": n, "minimal_samples": S, "hypotheses": real,
            "eager_latency_us": e_lat, "eager_pipelined_us": e_pipe, "graph_latency_us": g_lat, "graph_pipelined_us": g_pipe,
            "graph_latency_us_per_pair": g_lat / B, "graph_pairs_per_s_pipelined": B / (g_pipe * 1e-6),
            "graph_latency_us_resident_views": c_lat, "graph_pipelined_us_resident_views": c_pipe,
            "graph_table_equals_eager": same, "solver_scratch": "static (noalloc)" if p.noalloc else "workspace"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, nargs="*", default=[1, 8])
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    for B in args.pairs:
        print(json.dumps(measure(B, args.iters, dev)), flush=True)


if __name__ == "__main__":
    main()
