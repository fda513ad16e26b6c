#!/usr/bin/env python3
"""The multi-view configurations of BASELINE.json through the relative-pose pipeline.

    python benchmarks/configs.py --cfg 3            # 100-camera ring, 4,950 pairs, 32k binary descriptors
    python benchmarks/configs.py --cfg 4 --pairs N  # 1,000 views, 8k binary descriptors, first N of 499,500 pairs (0 = all)
    python benchmarks/configs.py --cfg 4 --f32      # the same with 8k x 128 float32 descriptors (L2 matcher)
    python benchmarks/configs.py --cfg 5            # 32 frames x 65,536 2-D/3-D correspondences, 16,384 P3P hypotheses

Prints one JSON line per run: pairs/s over the pair list (views resident, pairs in chunks that alternate over
--streams HIP streams, no host round trip inside a chunk), the accuracy against the synthetic ground truth and the
roofline of the run's dominant kernel.  `run_allpairs` / `run_cfg5` are what bench.py's `other_configs` calls.
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

VALU_PEAK = 256 * 4 * 32 * 2.4e9       # lane-ops/s (MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz)
MFMA_F16_PEAK = 2516.6e12
MFMA_FP4_PEAK = 10066.3e12             # block-scaled fp4 MFMA: 4 x the bf16 rate per clock


def _median_ms(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def pose_errors(tab, Rs, ts, vi, vj):
    """Rotation / translation-direction errors (degrees) of pose-table rows against the scene's ground truth, vectorised."""
    R1, R2 = Rs[vi], Rs[vj]
    R = R2 @ np.transpose(R1, (0, 2, 1))
    t = ts[vj] - np.einsum("nij,nj->ni", R, ts[vi])
    D = tab[:, :9].reshape(-1, 3, 3).astype(np.float64) @ np.transpose(R, (0, 2, 1))
    s = 0.5 * np.sqrt((D[:, 2, 1] - D[:, 1, 2]) ** 2 + (D[:, 0, 2] - D[:, 2, 0]) ** 2 + (D[:, 1, 0] - D[:, 0, 1]) ** 2)
    rerr = np.degrees(np.arctan2(s, (np.trace(D, axis1=1, axis2=2) - 1) / 2))
    a = tab[:, 9:12].astype(np.float64)
    terr = np.degrees(np.arctan2(np.linalg.norm(np.cross(a, t), axis=1), np.einsum("ni,ni->n", a, t)))
    return rerr, terr


def run_allpairs(cfg, f32, dev, n_pairs=0, samples=410, chunk=64, n_streams=3, views=0, feat=0, refine=True,
                 rank=0, world=1, dist=None, log=None):
    """cfg 3 / cfg 4 on this rank's shard (pair_id % world), ONE gather of the pose table at the end.  Returns the JSON
    dict (accuracy and roofline only on rank 0).  Scene generation and the upload of the views are outside `seconds`."""
    V = views or (100 if cfg == 3 else 1000)
    n = feat or (32768 if cfg == 3 else 8192)
    t0 = time.perf_counter()
    sc = (synthetic.float_views if f32 else synthetic.binary_views)(V, n, seed=1000 + cfg, config=cfg)
    gen_s = time.perf_counter() - t0
    if log:
        log(f"# rank {rank}: cfg {cfg} {'f32' if f32 else 'binary'} scene generated in {gen_s:.1f} s")
    desc = torch.from_numpy(sc["desc"]).to(dev)
    kp = torch.from_numpy(sc["kp"]).to(dev)
    norms = matching.row_sqnorm(desc) if f32 else None
    vi, vj = synthetic.all_pairs(V)
    total_pairs = len(vi)
    if n_pairs:
        vi, vj = vi[:n_pairs], vj[:n_pairs]
    thr_sq = float(np.float32((1.5 / synthetic.FX) ** 2))
    run = AllPairs(desc, kp, norms, thr_sq, samples, f32, refine=refine, resident_images=os.environ.get("POSEHOT_RESIDENT_IMAGES", "1") != "0")

    w = min(chunk, len(vi))                 # warm-up chunk (also the first real chunk's shapes), outside the timed region
    run.chunk(0, torch.arange(w), vi[:w], vj[:w]); run.reset_counts()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, n_streams))]

    def progress(k, done):
        if log and k % 256 == 255:
            log(f"# rank {rank}: {done} pairs enqueued")

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    full, local, mine = sharding.run_pair_list(vi, vj, run.chunk, rank, world, chunk=chunk,
                                               stream_ctx=lambda k: torch.cuda.stream(streams[k % len(streams)]),
                                               sync=torch.cuda.synchronize, progress=progress, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    real = run.real_models()
    if world > 1:
        red = torch.tensor([dt, 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(red[:1], op=dist.ReduceOp.MAX)
        cnt = torch.tensor([real], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt)
        dt, real = float(red[0].item()), int(cnt.item())
    line = {"cfg": cfg, "descriptors": "f32x128" if f32 else "256-bit", "streams": n_streams, "views": V, "features": n,
            "pairs": int(len(vi)), "pairs_of_the_scene": int(total_pairs), "pairs_this_rank": int(len(mine)), "rank": rank,
            "world": world, "seconds": dt, "pairs_per_s": len(vi) / dt,
            "minimal_samples": samples, "hypothesis_slots": 10 * samples,
            "hypotheses": real / len(vi), "hyps_per_s": real / dt, "scoring": "bail-out",
            "refine": "device LM, truncated Sampson cost (3 x threshold) over all matches" if refine else "none",
            "gather": "one all_gather_into_tensor of the [pairs,16] pose table after the last chunk, inside the timed region",
            "scene_generation_s": gen_s}
    if rank == 0:
        tab = full.cpu().numpy()
        assert tab.shape[0] == len(vi)
        rerr, terr = pose_errors(tab, sc["R"], sc["t"], vi, vj)
        line.update({"median_matches": float(np.median(tab[:, 13])), "median_inliers": float(np.median(tab[:, 12])),
                     "status_ok": float(np.mean(tab[:, 15] == pipeline.STATUS_OK)),
                     "rot_err_deg": {"median": float(np.median(rerr)), "p95": float(np.percentile(rerr, 95)), "max": float(rerr.max())},
                     "t_dir_err_deg": {"median": float(np.median(terr)), "p95": float(np.percentile(terr, 95)), "max": float(terr.max())},
                     "ok_within_1deg": float(np.mean((rerr < 1.0) & (terr < 1.0)))})
        # roofline of the run's dominant kernel: the matcher of ONE chunk, alone on the device (HIP events on its stream)
        B = min(chunk, len(vi))
        q = torch.as_tensor(np.ascontiguousarray(vi[:B]), dtype=torch.int32).to(dev)
        t = torch.as_tensor(np.ascontiguousarray(vj[:B]), dtype=torch.int32).to(dev)
        if f32:
            ms = _median_ms(lambda: matching.match_l2_ratio(desc, desc, 0.8, q, t, norm_q=norms, norm_t=norms))
            flops = 2.0 * n * n * 128 * B
            line["roofline"] = {"kernel": "ph_match_l2_ratio (prep + presplit + l2_filter_kernel<4> + l2_rerank_kernel<fused>)",
                                "bound": "mfma", "unit": "TFLOP/s", "achieved": flops / (ms * 1e-3) / 1e12,
                                "peak": MFMA_F16_PEAK / 1e12, "frac": flops / (ms * 1e-3) / MFMA_F16_PEAK,
                                "launch_ms": ms, "pairs_per_launch": B, "algorithmic_flops_per_launch": flops, "traffic": None}
        else:
            ms = _median_ms(lambda: matching.match_hamming(desc, desc, q, t))
            ops = 512.0 * n * n * B               # 256 fp4 multiply-adds per descriptor pair, as the kernel executes them
            line["roofline"] = {"kernel": "ph_match_hamming_top2 (hamming_expand_kernel + hamming_fp4_dma_kernel<2>)",
                                "bound": "mfma", "unit": "TOP/s", "achieved": ops / (ms * 1e-3) / 1e12,
                                "peak": MFMA_FP4_PEAK / 1e12, "frac": ops / (ms * 1e-3) / MFMA_FP4_PEAK,
                                "valu_equivalent_frac": 16.0 * n * n * B / (ms * 1e-3) / VALU_PEAK,
                                "valu_equivalent_note": "SURVEY.md §8d's count: 16 VALU lane-ops per descriptor pair against the "
                                                        "nominal VALU peak - the rate a popcount kernel would have to reach",
                                "launch_ms": ms, "pairs_per_launch": B, "algorithmic_ops_per_launch": ops, "traffic": None,
                                "share_of_run": ms * 1e-3 * (len(mine) / B) / dt}
    del desc, kp, norms, run, full, local
    torch.cuda.empty_cache()
    return line


def run_cfg5(dev, frames=32, N=65536, samples=0, iters=5):
    """PnP stress (SURVEY.md §8d cfg 5): per frame S Philox 3-point samples -> P3P (4 slots each) -> scoring at `thr` px ->
    winner -> mask -> LM on the inliers.  "16k hypotheses" are MODELS: S is chosen so that the mean number of real P3P
    solutions per frame (sum of n_solutions) reaches 16,384 (a slot holds a model only for a real solution).
    Returns one dict per threshold (8 px, 2 px)."""
    from pose_estimation_amd import board, ransac
    B = frames
    t0 = time.perf_counter()
    fr = synthetic.pnp_frames(B, N, outliers=0.3, noise_px=1.0, seed=1005)
    gen_s = time.perf_counter() - t0
    pts = [torch.from_numpy(fr[k]).to(dev) for k in "XYZuv"]
    K4 = torch.tensor([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY], dtype=torch.float64, device=dev)
    if samples:
        S = samples
    else:                       # probe the solutions-per-sample rate of this scene, then size S for >= 16,384 real models
        probe = pipeline.absolute_pose_batch(*pts, K4, thr_px=8.0, n_samples=1024, seed=0, refine=False, want_counts=True)
        per_sample = float(probe["n_solutions"].float().mean().item())
        S = int(np.ceil(16384 / per_sample * 1.01 / 64) * 64)
    lines = []
    for thr in (8.0, 2.0):
        run = lambda seed: pipeline.absolute_pose_batch(*pts, K4, thr_px=thr, n_samples=S, seed=seed, max_iters=20, want_counts=True)
        run(0); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        ev[0].record()
        for k in range(iters):
            out = run(k + 1)
            ev[k + 1].record()
        torch.cuda.synchronize()
        ms = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(iters)])
        real = float(out["n_solutions"].sum(dim=1).float().mean().item())
        # scoring alone, same hypotheses: exhaustive (the kernel SURVEY.md §8d prices) and with bail-out (what the batch runs)
        P = out["P_compact"]          # the live models packed: what the pipeline scores
        ms_ex = _median_ms(lambda: ransac.score_pnp(P, *pts, thr * thr), iters=3, warmup=1)
        ms_bo = _median_ms(lambda: ransac.score_pnp(P, *pts, thr * thr, bailout=True), iters=3, warmup=1)
        pose = out["pose"].cpu().numpy()
        rerr = np.array([geometry.rotation_angle_deg(board.rodrigues(pose[b, :3]), fr["R"][b]) for b in range(B)])
        terr = np.array([np.linalg.norm(pose[b, 3:] - fr["t"][b]) for b in range(B)])
        bc = out["best_count"].cpu().numpy()
        lane_ops = 17.0 * B * real * N            # SURVEY.md §8d / DESIGN §3: 17 lane-ops per (hypothesis, correspondence)
        lines.append({"cfg": 5, "frames": B, "correspondences": N, "minimal_samples": S, "hypothesis_slots": 4 * S,
                      "hypotheses": real, "threshold_px": thr, "scoring": "bail-out",
                      "ms_per_batch": float(np.median(ms)), "frames_per_s": B / (np.median(ms) * 1e-3),
                      "hyps_per_s": B * real / (np.median(ms) * 1e-3), "score_kernel_ms_exhaustive": ms_ex,
                      "score_kernel_ms_bailout": ms_bo,
                      "score_only_hyps_per_s_exhaustive": B * real / (ms_ex * 1e-3),
                      "roofline": {"kernel": "score_kernel<Pnp> (exhaustive, the live models)", "bound": "valu", "unit": "lane-ops/s",
                                   "achieved": lane_ops / (ms_ex * 1e-3), "peak": VALU_PEAK,
                                   "frac": lane_ops / (ms_ex * 1e-3) / VALU_PEAK, "launch_ms": ms_ex,
                                   "algorithmic_lane_ops_per_launch": lane_ops, "traffic": None,
                                   "bailout_launch_ms": ms_bo, "bailout_share_of_batch": ms_bo / float(np.median(ms))},
                      "scene_generation_s": gen_s, "median_inliers": float(np.median(bc)),
                      "true_inliers": float(fr["inlier_truth"].sum(1).mean()),
                      "rot_err_deg": {"median": float(np.median(rerr)), "max": float(rerr.max())},
                      "t_err_m": {"median": float(np.median(terr)), "max": float(terr.max())},
                      "rms_px": float(out["rms"].median().item()), "lm_iters": int(out["iters"].max().item())})
    del pts
    torch.cuda.empty_cache()
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=3)
    ap.add_argument("--views", type=int, default=0)
    ap.add_argument("--feat", type=int, default=0)
    ap.add_argument("--pairs", type=int, default=0, help="limit the pair list (0 = all)")
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--samples", type=int, default=410, help="minimal samples per pair (920 = the headline's >= 4096 real hypotheses); cfg 5: 0 = sized for 16,384 real models")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--f32", action="store_true", help="cfg 3/4 with float32 x 128 descriptors and the L2 matcher")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--no-refine", action="store_true", help="report the winning minimal-sample model as it is")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if args.cfg == 5:
        for line in run_cfg5(dev, args.frames, args.feat or 65536, 0 if args.samples == 410 else args.samples, args.iters):
            print(json.dumps(line), flush=True)
        return
    dist = None
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
    line = run_allpairs(args.cfg, args.f32, dev, args.pairs, args.samples, args.chunk, args.streams, args.views, args.feat,
                        not args.no_refine, rank, world, dist, log=lambda s: print(s, file=sys.stderr, flush=True))
    if rank != 0:
        print("# " + json.dumps({k: line[k] for k in ("rank", "pairs_this_rank", "seconds")}), file=sys.stderr, flush=True)
    else:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
