#!/usr/bin/env python3
"""bench.py — image pairs/s of the two-view relative-pose hot path on MI355X.

Workload (BASELINE.json configs[1]): synthetic two-view pairs, 8192 SIFT-style float32
descriptors (dim 128) per view, >= 4096 RANSAC essential-matrix hypotheses (REAL models: a 5-point
minimal sample yields ~4.6 real roots, so 920 samples = 9200 hypothesis slots) per pair.  One step =
one pass of the hot path over a batch of --pairs such pairs per GPU, inputs resident in HBM:

    L2 top-2 + ratio test (row norms inside the call; fp16 MFMA candidate filter + exact f32 re-rank) -> compaction -> Philox minimal samples ->
    5-point solver (hypothesis generation) -> Sampson scoring of the hypotheses (bail-out) -> best -> inlier mask ->
    decomposition of the winning E (closed form on the device; --decompose host = NumPy SVD) -> 4-candidate
    triangulation + cheirality vote -> pose table row -> LM refinement of the winning pose on the truncated Sampson cost

Every step's [pairs,16] pose table lands in a [steps*pairs,16] buffer; when N > 1 there is ONE collective,
an all_gather_into_tensor of that buffer after the last step, inside the timed region (north_star: final gather only).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 either launched under
torch.distributed.run (one rank per GPU, env RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*), or — when WORLD_SIZE is not
set — this script starts the N ranks itself as a child torch.distributed.run before touching the GPU and relays
rank 0's line.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2516.6   # MI355X_MICROARCH.md: ~2.5 PF dense f16/bf16 = 1024 SIMDs x 32768 FLOP / 32 cyc x 2.4 GHz
# HBM bytes of l2_filter_kernel alone come from the committed counter summary of `tools/gpu_pmc.sh` (rocprofv3 --pmc, separate
# passes for FETCH_SIZE and WRITE_SIZE; KiB per launch): FETCH_SIZE x 2 (the guide's gfx950 correction for 16-B-per-lane
# streaming reads) + WRITE_SIZE, scaled from the profile's pair count to this run's.  No file, no number.
FILTER_PMC_SUMMARY = ("profiles/r3/l2_matcher_pmc_r3_64pairs_summary.json", 64)       # (path under the repo root, pairs per launch)
N_FEAT, DIM = 8192, 128
N_SAMPLES = 920                 # minimal samples per pair; ~4.6 real essential matrices each => >= 4096 real hypotheses
THR_PX = 1.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps (200 x 1.8 ms: the first steps of a region fill the three-stream pipeline and the chip's "
                         "power management takes ~0.1 s to settle; 20 steps read 3-4 %% lower)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=64, help="pairs per step per GPU (64: the filter kernel runs 12 % faster per pair than at 32 — longer launches, smaller tails)")
    ap.add_argument("--samples", type=int, default=N_SAMPLES, help="5-point minimal samples per pair (10 hypothesis slots each)")
    ap.add_argument("--cpu-pairs", type=int, default=32, help="pairs timed through the CPU oracle (0 = skip)")
    ap.add_argument("--precision", type=int, default=1,
                    help="L2 matcher algorithm (identical bit-exact outputs): 0 = every score on f32 MFMA, "
                         "1 = fp16 MFMA candidate filter + exact f32 re-rank")
    ap.add_argument("--unfused", action="store_true",
                    help="ph_match_l2_top2 + ph_ratio_filter_l2 instead of ph_match_l2_ratio (same idx / keep)")
    ap.add_argument("--given-norms", action="store_true",
                    help="ph_row_sqnorm of both tables first and the norms handed to the matcher (the two-pass path of rounds 1-2) "
                         "instead of the matcher computing them in the passes that read the tables anyway")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the batches alternate over: with 2, the latency-bound RANSAC stages of one batch "
                         "run beside the MFMA-bound matcher of the next")
    ap.add_argument("--cu-split", type=int, default=0,
                    help="experiment (VERDICT r2 item 7): every batch slot gets TWO streams created with CU masks — the matcher on "
                         "256 - K CUs, the RANSAC tail on the other K — instead of one unmasked stream (0 = off; measured and "
                         "dropped: profiles/r3/cu_split.txt)")
    ap.add_argument("--decompose", default="device", choices=["device", "host"],
                    help="where the winning E is decomposed into its (R, t) candidates: closed form on the device "
                         "(no host round trip) or NumPy SVD on the host")
    ap.add_argument("--no-refine", action="store_true", help="skip the LM refinement of the winning pose")
    ap.add_argument("--h2d", action="store_true",
                    help="PCIe-inclusive variant (never the headline): every step first copies its batch (descriptors + "
                         "keypoints) from pinned host memory into a per-stream device buffer on the step's stream")
    ap.add_argument("--sustained-s", type=float, default=2.0,
                    help="seconds of back-to-back steps measured AFTER the timed region for the `sustained` key (0 = skip)")
    ap.add_argument("--check", action="store_true", help="print pose accuracy of the last step against ground truth (stderr)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` object (BASELINE.json configs[2..4], the single-pair latency of configs[1] and "
                         "the f32-only matcher's step rate; N = 1 only, ~40 s, outside the timed region)")
    ap.add_argument("--cfg4-pairs", type=int, default=100000, help="pairs of cfg 4's 499,500 that `other_configs` runs (0 = all)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a child
    torch.distributed.run BEFORE this process touches the GPU, relay rank 0's JSON line, exit with the child's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout.splitlines():
        if line.startswith("{"):
            print(line, flush=True)
        else:
            print(line, file=sys.stderr)
    raise SystemExit(proc.returncode)


class HipEvents:
    """Raw hipEvent_t pairs for ph_match_l2_top2_timed: the events are recorded by the library on the stream the
    dominant kernel is launched on (torch.cuda.Event only ever sees torch's current stream and cannot reach inside a call)."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def pair(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        for e in (a, b):
            rc = self.hip.hipEventCreate(ctypes.byref(e))
            if rc != 0:
                raise RuntimeError(f"hipEventCreate failed ({rc})")
        return a, b

    def elapsed_ms(self, pair):
        ms = ctypes.c_float()
        rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), pair[0], pair[1])
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed ({rc})")
        return float(ms.value)


def make_batch(pairs, rank, dev):
    import numpy as np
    import torch
    from pose_estimation_amd import synthetic
    d1, d2, k1, k2, truth = [], [], [], [], []
    for p in range(pairs):
        sc = synthetic.two_view_pair(n=N_FEAT, dim=DIM, seed=1002 + 7919 * rank + p)
        d1.append(sc["desc1"]); d2.append(sc["desc2"]); k1.append(sc["kp1"]); k2.append(sc["kp2"])
        truth.append((sc["R"], sc["t"]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.stack(a))).to(dev)
    return dict(desc_q=t(d1), desc_t=t(d2), kp_q=t(k1), kp_t=t(k2), truth=truth,
                host=dict(desc_q=d1, desc_t=d2, kp_q=k1, kp_t=k2))


def opencv_note():
    """the reference's CPU path is OpenCV; time it if it can be imported, say so if not."""
    try:
        import cv2  # noqa: F401
    except ImportError:
        return "OpenCV unavailable (import cv2: ModuleNotFoundError) - the reference's own CPU path cannot be timed on this box"
    return "OpenCV importable: " + cv2.__version__ + " (not timed: bench.py's CPU leg is the oracle port)"


def cpu_baseline(batch, n_pairs, threads, E_dev, real_models):
    """The CPU oracle ('port') over the same stages on a bounded sample of the same workload, and — beside it, as a
    second stated figure — the matcher alone as NumPy/OpenBLAS sgemm + top-2 (what a SIMD BFMatcher roughly costs).

    Hypothesis generation and the pose refinement are NOT part of the CPU leg (the oracle's 5-point solver and LM are
    NumPy and would dominate it): the CPU scores, exhaustively, the live models the GPU generated — the same set the GPU
    scores with bail-out.  OpenCV, the reference's own CPU path, is not installed on either box, so the
    `opencv` key says "unavailable" unless `import cv2` succeeds."""
    import numpy as np
    from oracle import oracle as orc
    from pose_estimation_amd import geometry, synthetic
    orc.set_num_threads(threads)
    thr_sq = np.float32((THR_PX / synthetic.FX) ** 2)
    h = dict(batch["host"], E=E_dev.cpu().numpy())
    t0 = time.perf_counter()
    t_match = 0.0
    for p in range(n_pairs):
        tm = time.perf_counter()
        idx, a, b = orc.match_l2_top2(h["desc_q"][p], h["desc_t"][p])
        t_match += time.perf_counter() - tm
        keep = orc.ratio_filter_l2(idx, a, b, 0.75)
        m = orc.compact_matches(idx, keep, h["kp_q"][p], h["kp_t"][p])
        n = m["n_match"]
        pts = [m[k][:n] for k in ("x1", "y1", "x2", "y2")]
        live = h["E"][p][np.any(h["E"][p] != 0.0, axis=1)]          # the models the GPU path scores (ph_compact_models)
        cnt = orc.score_essential(live, *pts, thr_sq)
        bi, _ = orc.select_best(cnt)
        mask = orc.inlier_mask_essential(live[bi], *pts, thr_sq, N_FEAT)
        P1, P2, _, _ = geometry.candidate_projections(live[bi].reshape(1, 3, 3))
        for c in range(4):
            orc.triangulate(P1[0, c], P2[0, c], *pts, mask)
    dt = time.perf_counter() - t0
    # second figure: BLAS matcher (not bit-exact with the contract; a fairer stand-in for OpenCV's SIMD BFMatcher)
    tb = time.perf_counter()
    nb = min(n_pairs, 8)
    for p in range(nb):
        q, t = h["desc_q"][p], h["desc_t"][p]
        s = q @ t.T
        s -= 0.5 * np.einsum("ij,ij->i", t, t)[None, :]
        top = np.argpartition(-s, 2, axis=1)[:, :2]
        _ = np.take_along_axis(s, top, axis=1)
    blas_dt = (time.perf_counter() - tb) / nb
    rest = (dt - t_match) / n_pairs
    return {"value": n_pairs / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{n_pairs} pair(s) of the same workload through oracle/posehot_oracle.c (OpenMP, {threads} threads), "
                      f"exhaustive scoring of the ~{real_models:.0f} live models per pair the GPU generated (empty solver slots skipped, "
                      f"as on the GPU), hypothesis generation and refinement excluded, {dt:.1f} s",
            "opencv": opencv_note(),
            "blas_matcher": {"value": 1.0 / (blas_dt + rest), "unit": "pairs/s",
                             "note": "same CPU leg with the matcher replaced by NumPy/OpenBLAS sgemm q@t.T + argpartition top-2 "
                                     f"({blas_dt * 1e3:.0f} ms per pair over {nb} pairs; NOT bit-exact with the contract's fma chain), "
                                     "the rest of the stages as in `value`"}}


def filter_traffic(pairs):
    """(HBM bytes of one l2_filter_kernel launch at `pairs` pairs, where the figure comes from) from the committed PMC summary."""
    path, prof_pairs = FILTER_PMC_SUMMARY
    try:
        d = json.load(open(os.path.join(ROOT, path)))
    except (OSError, ValueError):
        return None, f"{path} not found"
    for name, c in d.items():
        if "l2_filter_kernel" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            per_launch = (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0
            return per_launch * pairs / prof_pairs, f"{path}: (2 x FETCH_SIZE + WRITE_SIZE) KiB of l2_filter_kernel at {prof_pairs} pairs, scaled to {pairs}"
    return None, f"{path} holds no l2_filter_kernel counters"


def other_configs(args, dev, step_rate):
    """BASELINE.json's other configurations on this GPU, each with the roofline of its dominant kernel (driver-observed:
    part of the ONE JSON line).  Scene generation and uploads are outside every timed region."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    import configs as cf
    import latency
    log = lambda msg: print(msg, file=sys.stderr, flush=True)
    out = {"note": "N = 1, after the headline's timed region; pairs in chunks of 64 alternating over 3 HIP streams, views resident, "
                   f"{args.samples} minimal samples per pair (>= 4096 real hypotheses, as the headline), refinement on"}
    t0 = time.perf_counter()
    out["configs[2] cfg3: 100-camera ring, 4950 pairs, 32768 x 256-bit"] = cf.run_allpairs(3, False, dev, 0, args.samples, log=log)
    out["configs[3] cfg4 binary: 1000 views, 8192 x 256-bit"] = cf.run_allpairs(4, False, dev, args.cfg4_pairs, args.samples, log=log)
    out["configs[3] cfg4 f32: 1000 views, 8192 x 128 f32"] = cf.run_allpairs(4, True, dev, args.cfg4_pairs, args.samples, log=log)
    out["configs[4] cfg5: 32 frames x 65536 correspondences, >= 16384 P3P hypotheses"] = cf.run_cfg5(dev)
    out["configs[1] literally: ONE pair per batch (and 8), end to end"] = [latency.measure(B, 30, dev) for B in (1, 8)]
    out["f32_only_matcher"] = step_rate(0)
    out["seconds_spent"] = time.perf_counter() - t0
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)                                  # never returns
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    from pose_estimation_amd import matching, pipeline, synthetic

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
        os.environ.setdefault("MASTER_PORT", "29511")
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

    P, S = args.pairs, args.samples
    batch = make_batch(P, rank, dev)
    thr_sq = float(np.float32((THR_PX / synthetic.FX) ** 2))
    tables = torch.zeros((args.steps * P, 16), dtype=torch.float32, device=dev)            # this rank's rows, step after step
    gathered = torch.empty((world * args.steps * P, 16), dtype=torch.float32, device=dev) if use_dist else None
    hev = HipEvents()
    kev = [hev.pair() for _ in range(args.steps)]          # around the dominant kernel of every timed step
    state = {"step": -1, "timed": False}
    counts = []                                            # per-step [P, S] numbers of real models (summed after the region)

    keys = ("desc_q", "desc_t", "kp_q", "kp_t")
    pinned = {k: batch[k].cpu().pin_memory() for k in keys} if args.h2d else None
    slot_bufs = [{k: torch.empty_like(batch[k]) for k in keys} for _ in range(max(args.streams, 1))] if args.h2d else None
    cur = {"buf": batch}

    fused = args.precision == 1 and not args.unfused
    mode = {"precision": args.precision, "fused": fused}

    def match():
        b = cur["buf"]
        # f32 kernel / --given-norms: ph_row_sqnorm first; default: NULL norms, computed inside the matcher call
        two_pass = args.given_norms or mode["precision"] == 0
        nq = matching.row_sqnorm(b["desc_q"]) if two_pass else None
        nt = matching.row_sqnorm(b["desc_t"]) if two_pass else None
        k = state["step"]
        if mode["fused"]:          # matching + ratio test in one call: same idx / keep, exact re-scores only where needed
            return (*matching.match_l2_ratio(b["desc_q"], b["desc_t"], 0.75, norm_q=nq, norm_t=nt,
                                             kernel_events=kev[k] if state["timed"] else None), "l2_keep")
        r = matching.match_l2(b["desc_q"], b["desc_t"], norm_q=nq, norm_t=nt, precision=mode["precision"],
                              kernel_events=kev[k] if state["timed"] else None)
        return (*r, "l2")

    def split_match(slot):
        """The matcher on the slot's matcher stream (its own CU partition), the caller's stream waits for it."""
        tail, ms = torch.cuda.current_stream(), match_streams[slot]
        ms.wait_stream(tail)
        with torch.cuda.stream(ms):
            r = match()
        for t in r[:-1]:
            t.record_stream(tail)
        tail.wait_stream(ms)
        return r

    def front(slot=0):
        if args.h2d:
            cur["buf"] = slot_bufs[slot]
            for k in keys:
                cur["buf"][k].copy_(pinned[k], non_blocking=True)
        b = cur["buf"]
        return pipeline.relative_pose_front((lambda: split_match(slot)) if match_streams else match, b["kp_q"], b["kp_t"], None, None, None, thr_sq,
                                            n_samples=S, seed=max(state["step"], 0) + 1000,
                                            decompose=args.decompose, want_counts=True, refine=not args.no_refine)

    def back(f, k, timed):
        out = pipeline.relative_pose_back(f)
        if timed:
            tables[k * P:(k + 1) * P].copy_(out["table"], non_blocking=True)
            counts.append(out["n_solutions"])
        return out

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else [torch.cuda.current_stream()]
    main_stream = torch.cuda.current_stream()
    match_streams = None
    if args.cu_split > 0:
        # partitioned instead of shared CUs: stream pairs with complementary CU masks (hipExtStreamCreateWithCUMask)
        K = args.cu_split
        tail_bits = set(range(K))
        def masked_stream(bits):
            words = (ctypes.c_uint32 * 8)(*[sum(1 << (i % 32) for i in bits if i // 32 == w) for w in range(8)])
            h = ctypes.c_void_p()
            hev.hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
            rc = hev.hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
            if rc != 0:
                raise SystemExit(f"hipExtStreamCreateWithCUMask failed ({rc})")
            return torch.cuda.ExternalStream(h.value, device=dev)
        n_slots = max(args.streams, 1)
        streams = [masked_stream(tail_bits) for _ in range(n_slots)]
        match_streams = [masked_stream(set(range(256)) - tail_bits) for _ in range(n_slots)]
    for st in streams + (match_streams or []):
        st.wait_stream(main_stream)

    def run(n_steps, timed, gather=False):
        """n_steps whole batches.  The front half of batch k+1 is enqueued before the back half of batch k (with
        --decompose host the back half waits for a host round trip), every batch is completed (pose table row block
        written) inside the call.  Batch k lives on stream k mod --streams.  gather: after the last batch, the ONE
        collective of the job, on the main stream once every batch stream has been waited for."""
        pending, out = None, None
        state["timed"] = timed
        for k in range(n_steps):
            state["step"] = k
            st = streams[k % len(streams)]
            with torch.cuda.stream(st):
                f = front(k % len(streams))
            if pending is not None:
                with torch.cuda.stream(pending[1]):
                    out = back(pending[0], pending[2], timed)
            pending = (f, st, k)
        if pending is not None:
            with torch.cuda.stream(pending[1]):
                out = back(pending[0], pending[2], timed)
        if gather and use_dist:
            for st in streams:
                main_stream.wait_stream(st)
            dist.all_gather_into_tensor(gathered, tables)
        state["timed"] = False
        return out

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    out = run(args.steps, True, gather=True)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- what the timed region computed: real hypotheses per pair, matches per pair, the gathered table
    real_per_pair = float(torch.stack([c.sum(dim=1).float().mean() for c in counts]).mean().item())
    matches_per_pair = float(tables[:, 13].mean().item())
    inliers_per_pair = float(tables[:, 12].mean().item())
    if use_dist:
        stats = torch.tensor([real_per_pair, matches_per_pair, inliers_per_pair], dtype=torch.float64, device=dev)
        dist.all_reduce(stats)
        real_per_pair, matches_per_pair, inliers_per_pair = (stats / world).tolist()
        mine = gathered[rank * args.steps * P:(rank + 1) * args.steps * P]
        if not torch.equal(mine, tables):
            raise SystemExit("gathered pose table does not hold this rank's rows")
    if not bool((tables[:, 15] == pipeline.STATUS_OK).all()):
        raise SystemExit("a timed step left a pose-table row without STATUS_OK")

    # ---- sustained rate: back-to-back steps for >= --sustained-s seconds after the timed region (same code path)
    sustained = None
    if args.sustained_s > 0:
        n_sus = max(args.steps, int(args.sustained_s / (dt / args.steps)) + 1)
        barrier()
        ts = time.perf_counter()
        run(n_sus, False)
        barrier()
        sus_dt = time.perf_counter() - ts
        if use_dist:
            tmax = torch.tensor([sus_dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            sus_dt = float(tmax.item())
        sustained = {"value": world * P * n_sus / sus_dt, "unit": "pairs/s", "steps": n_sus, "seconds": sus_dt}

    def step_rate(precision, n_steps=12):
        """The same step with another matcher algorithm (identical outputs): pairs/s over n_steps untimed-region steps."""
        saved = dict(mode)
        mode.update(precision=precision, fused=False)
        try:
            run(3, False)
            barrier()
            ts = time.perf_counter()
            run(n_steps, False)
            barrier()
            d = time.perf_counter() - ts
        finally:
            mode.update(saved)
        return {"l2_precision": precision, "matcher": "ph_match_l2_top2 on f32 MFMA (l2_top2_f32_kernel<128>) + ph_ratio_filter_l2",
                "value": world * P * n_steps / d, "unit": "pairs/s", "steps": n_steps, "ms_per_step": d / n_steps * 1e3,
                "note": "the headline step with every score on the f32 matrix cores (157.3 TFLOP/s peak): same pose tables, bit for bit"}

    # ---- roofline of the dominant kernel.  frac = the kernel ALONE on the device (events recorded by the library around
    # the kernel, one stream, nothing else in flight): the number a rocprofv3 --kernel-trace of `bench.py --streams 1`
    # reproduces.  The brackets of the timed region (other streams' kernels share the CUs then) are kept as contended_*.
    flops = 2.0 * N_FEAT * N_FEAT * DIM * P
    contended_us = float(np.mean([hev.elapsed_ms(p) for p in kev])) * 1e3
    two_pass = args.given_norms or args.precision == 0
    nq_ = matching.row_sqnorm(batch["desc_q"]) if two_pass else None
    nt_ = matching.row_sqnorm(batch["desc_t"]) if two_pass else None
    iso_k, iso_call = [], []
    for _ in range(6):
        pr = hev.pair()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        if fused:
            matching.match_l2_ratio(batch["desc_q"], batch["desc_t"], 0.75, norm_q=nq_, norm_t=nt_, kernel_events=pr)
        else:
            matching.match_l2(batch["desc_q"], batch["desc_t"], norm_q=nq_, norm_t=nt_, precision=args.precision, kernel_events=pr)
        b.record()
        torch.cuda.synchronize()
        iso_k.append(hev.elapsed_ms(pr) * 1e3)
        iso_call.append(a.elapsed_time(b) * 1e3)
    kern_us, call_us = float(np.mean(iso_k[1:])), float(np.mean(iso_call[1:]))
    peak = PEAK_F32_MFMA_TFLOPS if args.precision == 0 else PEAK_F16_MFMA_TFLOPS
    achieved = flops / (kern_us * 1e-6) / 1e12
    traffic, traffic_src = filter_traffic(P) if args.precision == 1 else (None, "not profiled for --precision 0")
    roofline = {
        "kernel": "l2_top2_f32_kernel<128>" if args.precision == 0 else "l2_filter_kernel<4>",
        "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
        "traffic": traffic, "traffic_source": traffic_src,
        "dominant_kernel_us": kern_us,
        "how": "HIP events recorded by ph_match_l2_top2_timed around the kernel alone, on its stream, mean of 5 launches "
               "with nothing else in flight; reproduce: rocprofv3 --kernel-trace --stats -- python3 bench.py --streams 1",
        "algorithmic_flops_per_launch": flops,
        "executed_mfma_flops_per_launch": flops * (1 + 8 / 256) if args.precision == 1 else flops,
        "contended_kernel_us": contended_us, "contended_frac": flops / (contended_us * 1e-6) / 1e12 / peak,
        "contended_note": f"the same kernel inside the timed region, {args.streams} stream(s): other batches' kernels share the CUs",
        "matcher_call_us": call_us, "matcher_call_frac": flops / (call_us * 1e-6) / 1e12 / peak,
        "matcher_call": ("ph_match_l2_ratio" if fused else "ph_match_l2_top2") +
                        (" = l2_filter_prep_kernel + l2_presplit_kernel + l2_filter_kernel<4> + l2_rerank_kernel (norms handed in)" if two_pass else
                         " = l2_presplit_norm_kernel + l2_presplit_kernel<fix-up> + l2_filter_kernel<4> + l2_rerank_kernel (row norms computed inside)")
                        if args.precision == 1 else "ph_match_l2_top2 = l2_top2_f32_kernel<128>",
        "achieved_over_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
        "note": "outputs are bit-identical to the f32 MFMA matcher (--precision 0)"}

    if rank == 0 and args.precision == 1 and not args.no_other_configs:      # (profiled runs pass --no-other-configs: their kernel
                                                                             # statistics then hold the step's kernels only)
        # Context for `frac`, not a target: what the vendor's GEMM reaches on THIS box at a far easier shape (hipBLASLt through
        # torch.matmul, fp16, 8192^3, operands resident, the product written once) — the chip's fp16 matrix rate under its
        # power limit, against the nominal peak the fraction is taken over.
        ga = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
        gb = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
        for _ in range(3):
            ga @ gb
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ga @ gb
        e1.record()
        torch.cuda.synchronize()
        gemm_tf = 2.0 * 8192 ** 3 / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12
        roofline["vendor_gemm_same_box"] = {"what": "torch.matmul (hipBLASLt) fp16 8192 x 8192 x 8192, mean of 10", "achieved": gemm_tf,
                                            "frac_of_peak": gemm_tf / peak, "filter_over_vendor_gemm": achieved / gemm_tf}
        del ga, gb

    if args.check and rank == 0:
        from pose_estimation_amd import geometry
        tab = out["table"].cpu().numpy()
        for p, (R, t) in enumerate(batch["truth"]):
            ra = geometry.rotation_angle_deg(tab[p, :9].reshape(3, 3), R)
            ta = geometry.direction_angle_deg(tab[p, 9:12], t)
            print(f"# pair {p}: matches {int(tab[p,13])} inliers {int(tab[p,12])} rot err {ra:.4f} deg, t dir err {ta:.4f} deg",
                  file=sys.stderr)

    if rank == 0:
        pairs_s = world * P * args.steps / dt
        res = {
            "metric": "image pairs/sec (relative pose, whole node) at 8k feats; RANSAC hyps/sec",
            "value": pairs_s, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (outputs bit-exact f32; contraction on fp16 MFMA + exact f32 re-rank)" if args.precision == 1 else "f32",
            "data": "synthetic", "l2_precision": args.precision,
            "config": {"workload": "configs[1]: two-view pair, 8192 SIFT-style f32x128 descriptors per view, "
                                   ">= 4096 essential-matrix hypotheses (real models) per pair", "pairs_per_step_per_gpu": P,
                       "features": N_FEAT, "dim": DIM, "hypotheses": real_per_pair, "hypothesis_slots": 10 * S,
                       "minimal_samples": S, "ratio": 0.75, "row_norms": "ph_row_sqnorm, two-pass" if (args.given_norms or args.precision == 0) else "inside the matcher call (one read of each table)", "matcher": "ph_match_l2_ratio (matching + ratio test, exact re-scores only where the approximate scores do not decide)" if fused else "ph_match_l2_top2 + ph_ratio_filter_l2", "threshold_px": THR_PX, "scoring": "bail-out",
                       "refine": "none" if args.no_refine else f"device LM, {pipeline.REFINE_ITERS} iterations, Sampson cost truncated at {pipeline.REFINE_TRUNC:g} x threshold",
                       "streams": args.streams, "cu_split": args.cu_split or None, "decompose": args.decompose,
                       "inputs": "pinned host memory, copied over PCIe inside every step (NOT the headline)" if args.h2d else "resident in HBM",
                       "hypothesis_source": "generated in the timed region: Philox minimal samples of the pair's own matches -> device 5-point solver "
                                            "(10 slots per sample; `hypotheses` = mean number of REAL models per pair = sum of n_solutions)",
                       "parallelism": f"pairs sharded over {world} rank(s), ONE all_gather of the [steps*pairs,16] pose table after the last step"},
            "hyps_per_sec": pairs_s * real_per_pair, "hyps_per_sec_counts": "real models generated + scored (bail-out) per second; "
                            "exhaustive scoring alone: benchmarks/kernels.py, profiles/r2/kernels_*.jsonl",
            "matches_per_pair": matches_per_pair, "inliers_per_pair": inliers_per_pair,
            "sustained": sustained,
            "roofline": roofline,
        }
        if real_per_pair < 4096:
            res["config"]["warning"] = f"only {real_per_pair:.0f} real hypotheses per pair: raise --samples"
        if world == 1 and args.cpu_pairs > 0:
            threads = min(16, os.cpu_count() or 1)
            res["cpu_baseline"] = cpu_baseline(batch, min(args.cpu_pairs, P), threads, out["E_hyp"], real_per_pair)
        elif world > 1:
            res["cpu_baseline"] = "skipped (world > 1): the CPU leg is timed on rank 0 at N = 1 only"
        if world == 1 and not use_dist and not args.no_other_configs:
            res["other_configs"] = other_configs(args, dev, step_rate)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
