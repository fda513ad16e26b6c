#!/usr/bin/env python3
"""bench.py — image pairs/s of the two-view relative-pose hot path on MI355X.

Workload (BASELINE.json configs[1]): synthetic two-view pairs, 8192 SIFT-style float32
descriptors (dim 128) per view, 4096 RANSAC essential-matrix hypotheses per pair.  One step =
one pass of the hot path over a batch of --pairs such pairs per GPU, inputs resident in HBM:

    row norms -> L2 top-2 (fp16 MFMA candidate filter + exact f32 re-rank) -> ratio test -> compaction -> Philox minimal samples ->
    5-point solver (hypothesis generation) -> Sampson scoring of H hypotheses -> best -> inlier mask ->
    decomposition of the winning E (closed form on the device; --decompose host = NumPy SVD) -> 4-candidate
    triangulation + cheirality vote -> pose table  (-> all_gather of the table when N > 1)

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
torch.distributed.run, one rank per GPU; prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pose_estimation_amd import matching, pipeline, synthetic  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2516.6   # MI355X_MICROARCH.md: ~2.5 PF dense f16/bf16 = 1024 SIMDs x 32768 FLOP / 32 cyc x 2.4 GHz
L2_CALL_HBM_BYTES_AT_32_PAIRS = 816.0e6   # (1.1 + 66.3 + 100.0 + 168.8) MB FETCH_SIZE x 2 + (2.1 + 67.6 + 71.3 + 3.2) MB WRITE_SIZE
N_FEAT, DIM = 8192, 128
N_SAMPLES = 410                 # minimal samples per pair; the 5-point solver fills 10 hypothesis slots each
N_HYP = 10 * N_SAMPLES          # 4100 essential-matrix hypotheses scored per pair (>= the 4096 of configs[1])
THR_PX = 1.0


def make_batch(pairs, rank, dev):
    d1, d2, k1, k2, truth = [], [], [], [], []
    for p in range(pairs):
        sc = synthetic.two_view_pair(n=N_FEAT, dim=DIM, seed=1002 + 7919 * rank + p)
        d1.append(sc["desc1"]); d2.append(sc["desc2"]); k1.append(sc["kp1"]); k2.append(sc["kp2"])
        truth.append((sc["R"], sc["t"]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack(a))).to(dev)
    return dict(desc_q=t(d1), desc_t=t(d2), kp_q=t(k1), kp_t=t(k2), truth=truth,
                host=dict(desc_q=d1, desc_t=d2, kp_q=k1, kp_t=k2))


def cpu_baseline(batch, n_pairs, threads, E_dev):
    """The CPU oracle ('port') over the same stages on a bounded sample of the same workload.

    Hypothesis generation is NOT part of the CPU leg (the oracle's 5-point solver is NumPy and would
    dominate it): the CPU scores the hypotheses the GPU generated, which only flatters the CPU."""
    from oracle import oracle as orc
    orc.set_num_threads(threads)
    thr_sq = np.float32((THR_PX / synthetic.FX) ** 2)
    h = dict(batch["host"], E=E_dev.cpu().numpy())
    t0 = time.perf_counter()
    for p in range(n_pairs):
        idx, a, b = orc.match_l2_top2(h["desc_q"][p], h["desc_t"][p])
        keep = orc.ratio_filter_l2(idx, a, b, 0.75)
        m = orc.compact_matches(idx, keep, h["kp_q"][p], h["kp_t"][p])
        n = m["n_match"]
        pts = [m[k][:n] for k in ("x1", "y1", "x2", "y2")]
        cnt = orc.score_essential(h["E"][p], *pts, thr_sq)
        bi, _ = orc.select_best(cnt)
        mask = orc.inlier_mask_essential(h["E"][p][bi], *pts, thr_sq, N_FEAT)
        from pose_estimation_amd import geometry
        P1, P2, _, _ = geometry.candidate_projections(h["E"][p][bi].reshape(1, 3, 3))
        for c in range(4):
            orc.triangulate(P1[0, c], P2[0, c], *pts, mask)
    dt = time.perf_counter() - t0
    return {"value": n_pairs / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{n_pairs} pair(s) of the same workload through oracle/posehot_oracle.c (OpenMP, {threads} threads), hypothesis generation excluded, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=32, help="pairs per step per GPU")
    ap.add_argument("--cpu-pairs", type=int, default=32, help="pairs timed through the CPU oracle (0 = skip)")
    ap.add_argument("--precision", type=int, default=1,
                    help="L2 matcher algorithm (identical bit-exact outputs): 0 = every score on f32 MFMA, "
                         "1 = fp16 MFMA candidate filter + exact f32 re-rank")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the batches alternate over: with 2, the latency-bound RANSAC stages of one batch "
                         "run beside the MFMA-bound matcher of the next")
    ap.add_argument("--decompose", default="device", choices=["device", "host"],
                    help="where the winning E is decomposed into its (R, t) candidates: closed form on the device "
                         "(no host round trip) or NumPy SVD on the host")
    ap.add_argument("--h2d", action="store_true",
                    help="PCIe-inclusive variant (never the headline): every step first copies its batch (descriptors + "
                         "keypoints) from pinned host memory into a per-stream device buffer on the step's stream")
    ap.add_argument("--check", action="store_true", help="verify pose accuracy of the last step against ground truth")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # POSEHOT_BENCH_FORCE_DIST=1 takes the collective path at world size 1 too (rehearsal of the N > 1 code on a one-GPU box)
    use_dist = world > 1 or os.environ.get("POSEHOT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    P = args.pairs
    batch = make_batch(P, rank, dev)
    thr_sq = float(np.float32((THR_PX / synthetic.FX) ** 2))
    gathered = torch.empty((world * P, 16), dtype=torch.float32, device=dev) if use_dist else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    state = {"step": -1}

    keys = ("desc_q", "desc_t", "kp_q", "kp_t")
    pinned = {k: batch[k].cpu().pin_memory() for k in keys} if args.h2d else None
    slot_bufs = [{k: torch.empty_like(batch[k]) for k in keys} for _ in range(max(args.streams, 1))] if args.h2d else None
    cur = {"buf": batch}

    def match():
        b = cur["buf"]
        nq = matching.row_sqnorm(b["desc_q"])
        nt = matching.row_sqnorm(b["desc_t"])
        k = state["step"]
        if k >= 0:
            ev[k][0].record()
        r = matching.match_l2(b["desc_q"], b["desc_t"], norm_q=nq, norm_t=nt, precision=args.precision)
        if k >= 0:
            ev[k][1].record()
        return (*r, "l2")

    def front(slot=0):
        if args.h2d:
            cur["buf"] = slot_bufs[slot]
            for k in keys:
                cur["buf"][k].copy_(pinned[k], non_blocking=True)
        b = cur["buf"]
        return pipeline.relative_pose_front(match, b["kp_q"], b["kp_t"], None, None, None, thr_sq,
                                            n_samples=N_SAMPLES, seed=max(state["step"], 0) + 1000,
                                            decompose=args.decompose)

    def back(f):
        out = pipeline.relative_pose_back(f)
        if use_dist:
            dist.all_gather_into_tensor(gathered, out["table"])
        return out

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else [torch.cuda.current_stream()]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())

    def run(n_steps, timed):
        """n_steps whole batches.  The front half of batch k+1 is enqueued before the host finishes batch k
        (its 3x3 SVDs), so the GPU does not idle during the host round trip; every batch is completed
        (pose table written, gathered) inside the call.  Batch k lives on stream k mod --streams."""
        pending, out = None, None
        for k in range(n_steps):
            state["step"] = k if timed else -1
            st = streams[k % len(streams)]
            with torch.cuda.stream(st):
                f = front(k % len(streams))
            if pending is not None:
                with torch.cuda.stream(pending[1]):
                    out = back(pending[0])
            pending = (f, st)
        if pending is not None:
            with torch.cuda.stream(pending[1]):
                out = back(pending[0])
        return out

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    out = run(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    state["step"] = -1
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    flops = 2.0 * N_FEAT * N_FEAT * DIM * P
    achieved = flops / (kern_ms * 1e-3) / 1e12
    # the same matcher call alone on the device (no other stream's kernels beside it): 5 launches after the timed region
    nq_, nt_ = matching.row_sqnorm(batch["desc_q"]), matching.row_sqnorm(batch["desc_t"])
    iso = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        matching.match_l2(batch["desc_q"], batch["desc_t"], norm_q=nq_, norm_t=nt_, precision=args.precision)
        b.record()
        torch.cuda.synchronize()
        iso.append(a.elapsed_time(b))
    iso_ms = float(np.mean(iso[1:]))

    if args.check and rank == 0:
        from pose_estimation_amd import geometry
        tab = out["table"].cpu().numpy()
        for p, (R, t) in enumerate(batch["truth"]):
            ra = geometry.rotation_angle_deg(tab[p, :9].reshape(3, 3), R)
            ta = geometry.direction_angle_deg(tab[p, 9:12], t)
            print(f"# pair {p}: matches {int(tab[p,13])} inliers {int(tab[p,12])} rot err {ra:.4f} deg, t dir err {ta:.4f} deg",
                  file=sys.stderr)

    if args.precision == 0:
        roofline = {"kernel": "l2_top2_f32_kernel<128>", "bound": "mfma", "achieved": achieved,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                    "traffic": None, "launch_ms": kern_ms, "algorithmic_flops_per_launch": flops}
    else:
        # One ph_match_l2_top2 call = l2_filter_prep_kernel + l2_presplit_kernel + l2_filter_kernel<4> + l2_rerank_kernel; the
        # events bracket the call on its own stream.  `achieved` counts the ALGORITHMIC 2*N*M*D flops per pair; the filter
        # executes one fp16 MFMA product per algorithmic product (+3 % priming).  With --streams > 1 other batches' kernels
        # share the device during the call, so launch_ms is the duration under that contention; isolated_launch_ms is the
        # same call alone.
        roofline = {"kernel": "ph_match_l2_top2 = l2_filter_prep_kernel + l2_presplit_kernel + l2_filter_kernel<4> + l2_rerank_kernel",
                    "bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_F16_MFMA_TFLOPS,
                    # HBM bytes per launch, rocprofv3 --pmc offline (profiles/r1/l2_filter_f16_v9_traffic_pmc_summary.json): FETCH_SIZE x2
                    # (gfx950 correction) + WRITE_SIZE over the four kernels of the call at 32 pairs; scaled by pairs
                    "traffic": L2_CALL_HBM_BYTES_AT_32_PAIRS * P / 32, "launch_ms": kern_ms, "isolated_launch_ms": iso_ms,
                    "isolated_achieved": flops / (iso_ms * 1e-3) / 1e12,
                    "isolated_frac": flops / (iso_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
                    "algorithmic_flops_per_launch": flops, "executed_f16_mfma_flops_per_launch": flops * (1 + 8 / 256),
                    "achieved_over_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                    "note": "outputs are bit-identical to the f32 MFMA matcher (--precision 0, 81 % of the f32 MFMA peak)"}
    if rank == 0:
        res = {
            "metric": "image pairs/sec (relative pose, whole node) at 8k feats",
            "value": world * P * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "l2_precision": args.precision,
            "config": {"workload": "configs[1]: two-view pair, 8192 SIFT-style f32x128 descriptors per view, "
                                   "4096 essential-matrix hypotheses per pair", "pairs_per_step_per_gpu": P,
                       "features": N_FEAT, "dim": DIM, "hypotheses": N_HYP, "ratio": 0.75, "threshold_px": THR_PX,
                       "minimal_samples": N_SAMPLES, "streams": args.streams, "decompose": args.decompose,
                       "inputs": "pinned host memory, copied over PCIe inside every step (NOT the headline)" if args.h2d else "resident in HBM",
                       "hypothesis_source": "generated in the timed region: Philox minimal samples of the pair's own matches -> device 5-point solver (10 slots per sample)",
                       "parallelism": f"pairs sharded over {world} rank(s), final all_gather of the [pairs,16] pose table"},
            "hyps_per_sec": world * P * N_HYP * args.steps / dt,
            "roofline": roofline,
        }
        if world == 1 and args.cpu_pairs > 0:
            threads = min(16, os.cpu_count() or 1)
            res["cpu_baseline"] = cpu_baseline(batch, min(args.cpu_pairs, P), threads, out["E_hyp"])
        print(json.dumps(res))
    if use_dist:
        if rank == 0 and not torch.equal(gathered[:P], out["table"]):
            raise SystemExit("gathered pose table does not hold this rank's rows")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
