#!/usr/bin/env python3
"""The multi-view configurations of BASELINE.json through the relative-pose pipeline.

    python benchmarks/configs.py --cfg 3            # 100-camera ring, 4,950 pairs, 32k binary descriptors
    python benchmarks/configs.py --cfg 4 --pairs N  # 1,000 views, 8k binary descriptors, first N of 499,500 pairs
    python benchmarks/configs.py --cfg 4 --f32      # the same with 8k x 128 float32 descriptors (L2 matcher)
    python benchmarks/configs.py --cfg 5            # 32 frames x 65,536 2-D/3-D correspondences, 16,384 P3P hypotheses

Prints one JSON line: pairs/s over the whole pair list (views resident, pairs in chunks that alternate over
--streams HIP streams, no host round trip inside a chunk) and the accuracy against the synthetic ground truth.
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
from pose_estimation_amd.allpairs import AllPairs  # noqa: E402


def cfg5(args, dev):
    """PnP stress (SURVEY.md §8d cfg 5): per frame S Philox 3-point samples -> P3P (4 slots each) -> scoring at `thr` px ->
    winner -> mask -> LM on the inliers.  "16k hypotheses" are MODELS: S is chosen so that the mean number of real P3P
    solutions per frame (sum of n_solutions) reaches 16,384 (a slot holds a model only for a real solution).
    One JSON line per threshold."""
    from pose_estimation_amd import board, ransac
    B, N = args.frames, args.feat or 65536
    t0 = time.perf_counter()
    fr = synthetic.pnp_frames(B, N, outliers=0.3, noise_px=1.0, seed=1005)
    gen_s = time.perf_counter() - t0
    pts = [torch.from_numpy(fr[k]).to(dev) for k in "XYZuv"]
    K4 = torch.tensor([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY], dtype=torch.float64, device=dev)
    if args.samples != 410:
        S = args.samples
    else:                       # probe the solutions-per-sample rate of this scene, then size S for >= 16,384 real models
        probe = pipeline.absolute_pose_batch(*pts, K4, thr_px=8.0, n_samples=1024, seed=0, refine=False, want_counts=True)
        per_sample = float(probe["n_solutions"].float().mean().item())
        S = int(np.ceil(16384 / per_sample * 1.01 / 64) * 64)
    for thr in (8.0, 2.0):
        run = lambda seed: pipeline.absolute_pose_batch(*pts, K4, thr_px=thr, n_samples=S, seed=seed, max_iters=20, want_counts=True)
        run(0); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
        ev[0].record()
        for k in range(args.iters):
            out = run(k + 1)
            ev[k + 1].record()
        torch.cuda.synchronize()
        ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(args.iters)])
        real = float(out["n_solutions"].sum(dim=1).float().mean().item())
        # scoring alone, same hypotheses, exhaustive (the kernel SURVEY.md §8d prices): HIP events around the launch
        P = out["P_compact"]          # the live models packed: what the pipeline scores
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ransac.score_pnp(P, *pts, thr * thr); e1.record(); torch.cuda.synchronize()
        pose = out["pose"].cpu().numpy()
        rerr = np.array([geometry.rotation_angle_deg(board.rodrigues(pose[b, :3]), fr["R"][b]) for b in range(B)])
        terr = np.array([np.linalg.norm(pose[b, 3:] - fr["t"][b]) for b in range(B)])
        bc = out["best_count"].cpu().numpy()
        print(json.dumps({"cfg": 5, "frames": B, "correspondences": N, "minimal_samples": S, "hypothesis_slots": 4 * S,
                          "hypotheses": real, "threshold_px": thr, "scoring": "bail-out",
                          "ms_per_batch": float(np.median(ms)), "frames_per_s": B / (np.median(ms) * 1e-3),
                          "hyps_per_s": B * real / (np.median(ms) * 1e-3), "score_kernel_ms_exhaustive": e0.elapsed_time(e1),
                          "score_only_hyps_per_s_exhaustive": B * real / (e0.elapsed_time(e1) * 1e-3),
                          "scene_generation_s": gen_s, "median_inliers": float(np.median(bc)),
                          "true_inliers": float(fr["inlier_truth"].sum(1).mean()),
                          "rot_err_deg": {"median": float(np.median(rerr)), "max": float(rerr.max())},
                          "t_err_m": {"median": float(np.median(terr)), "max": float(terr.max())},
                          "rms_px": float(out["rms"].median().item()), "lm_iters": int(out["iters"].max().item())}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=3)
    ap.add_argument("--views", type=int, default=0)
    ap.add_argument("--feat", type=int, default=0)
    ap.add_argument("--pairs", type=int, default=0, help="limit the pair list (0 = all)")
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--samples", type=int, default=410)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--f32", action="store_true", help="cfg 3/4 with float32 x 128 descriptors and the L2 matcher")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--no-refine", action="store_true", help="report the winning minimal-sample model as it is")
    args = ap.parse_args()
    if args.cfg == 5:
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        return cfg5(args, dev)
    V = args.views or (100 if args.cfg == 3 else 1000)
    n = args.feat or (32768 if args.cfg == 3 else 8192)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)                       # RCCL's banner goes to stderr: stdout carries the ONE JSON line of rank 0
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    t0 = time.perf_counter()
    if args.f32:
        sc = synthetic.float_views(V, n, seed=1000 + args.cfg, config=args.cfg)
    else:
        sc = synthetic.binary_views(V, n, seed=1000 + args.cfg, config=args.cfg)
    gen_s = time.perf_counter() - t0
    print(f"# rank {rank}: scene generated in {gen_s:.1f} s", file=sys.stderr, flush=True)
    desc = torch.from_numpy(sc["desc"]).to(dev)
    kp = torch.from_numpy(sc["kp"]).to(dev)
    norms = matching.row_sqnorm(desc) if args.f32 else None
    vi, vj = synthetic.all_pairs(V)
    if args.pairs:
        vi, vj = vi[:args.pairs], vj[:args.pairs]
    thr_sq = float(np.float32((1.5 / synthetic.FX) ** 2))
    run = AllPairs(desc, kp, norms, thr_sq, args.samples, args.f32, refine=not args.no_refine)

    # warm-up chunk (also the first real chunk's shapes), outside the timed region
    w = min(args.chunk, len(vi))
    run.chunk(0, torch.arange(w), vi[:w], vj[:w]); run.real_models.zero_()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    def progress(k, done):
        if k % 256 == 255:
            print(f"# rank {rank}: {done} pairs enqueued", file=sys.stderr, flush=True)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # shard (pair_id % world), chunks alternating over the streams, ONE all_gather of the pose table at the end
    full, local, mine = sharding.run_pair_list(vi, vj, run.chunk, rank, world, chunk=args.chunk,
                                               stream_ctx=lambda k: torch.cuda.stream(streams[k % len(streams)]),
                                               sync=torch.cuda.synchronize, progress=progress, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    real = run.real_models.clone()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(real)
        dt = float(tmax.item())
    line = {"cfg": args.cfg, "descriptors": "f32x128" if args.f32 else "256-bit", "streams": args.streams,
            "views": V, "features": n, "pairs": int(len(vi)), "pairs_this_rank": int(len(mine)), "rank": rank, "world": world,
            "seconds": dt, "pairs_per_s": len(vi) / dt,
            "minimal_samples": args.samples, "hypothesis_slots": 10 * args.samples,
            "hypotheses": float(real.item()) / len(vi), "hyps_per_s": float(real.item()) / dt, "scoring": "bail-out",
            "refine": "device LM, truncated Sampson cost (3 x threshold) over all matches" if not args.no_refine else "none",
            "gather": "one all_gather_into_tensor of the [pairs,16] pose table after the last chunk, inside the timed region",
            "scene_generation_s": gen_s}
    if rank != 0:
        print("# " + json.dumps({k: line[k] for k in ("rank", "pairs_this_rank", "seconds")}), file=sys.stderr, flush=True)
    else:
        tab = full.cpu().numpy()
        assert tab.shape[0] == len(vi)
        rerr, terr = [], []
        for p in range(len(vi)):
            R, t = synthetic.relative_pose(sc["R"][vi[p]], sc["t"][vi[p]], sc["R"][vj[p]], sc["t"][vj[p]])
            rerr.append(geometry.rotation_angle_deg(tab[p, :9].reshape(3, 3), R))
            terr.append(geometry.direction_angle_deg(tab[p, 9:12], t))
        rerr, terr = np.array(rerr), np.array(terr)
        line.update({"median_matches": float(np.median(tab[:, 13])), "median_inliers": float(np.median(tab[:, 12])),
                     "rot_err_deg": {"median": float(np.median(rerr)), "p95": float(np.percentile(rerr, 95)), "max": float(rerr.max())},
                     "t_dir_err_deg": {"median": float(np.median(terr)), "p95": float(np.percentile(terr, 95)), "max": float(terr.max())},
                     "ok_within_1deg": float(np.mean((rerr < 1.0) & (terr < 1.0)))})
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
