#!/usr/bin/env python3
"""Per-kernel roofline microbench: each hot-path kernel at the BASELINE.json shape it is specified
for, timed with HIP events on the launch stream, against the peak that bounds it.

    python benchmarks/kernels.py [--iters 20] [--only l2,hamming,...]  -> one JSON line per kernel

Peaks (MI355X_MICROARCH.md): f32 MFMA 157.3 TFLOP/s; VALU 256 CU x 4 SIMD x 32 lanes x 2.4 GHz =
7.86e13 lane-ops/s (an fma counts as one lane-op); HBM 8 TB/s.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose_estimation_amd import matching, ransac, synthetic, triangulation  # noqa: E402

VALU_PEAK = 256 * 4 * 32 * 2.4e9
MFMA_F32_PEAK = 157.3e12
MFMA_F16_PEAK = 2516.6e12   # ~2.5 PF dense nominal (2.4 GHz); 1.25-1.5 PF sustained on random data (MI355X_MICROARCH.md)
MFMA_I8_PEAK = 5033.2e12     # i8 MFMA: 2x the bf16 rate per clock
MFMA_FP4_PEAK = 10066.3e12   # block-scaled fp4 MFMA: 4x the bf16 rate per clock (~10 PF dense)
HBM_PEAK = 8.0e12


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def report(name, shape, med, best, work, unit, peak, extra=None):
    r = {"kernel": name, "shape": shape, "median_ms": med * 1e3, "min_ms": best * 1e3, "work": work, "work_unit": unit,
         "achieved": work / med, "peak": peak, "frac": work / med / peak}
    if extra:
        r.update(extra)
    print(json.dumps(r), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    if not only or "l2" in only:
        for P in (16, 1):
            V, n, d = 2 * P, 8192, 128
            tab = torch.randn((V, n, d), device=dev).abs_()
            tab /= tab.norm(dim=2, keepdim=True)
            vq = torch.arange(0, V, 2, dtype=torch.int32, device=dev); vt = vq + 1
            nrm = matching.row_sqnorm(tab)
            med, best = timeit(lambda: matching.match_l2(tab, tab, vq, vt, norm_q=nrm, norm_t=nrm, precision=0), args.iters)
            report("l2_top2_f32_kernel<128>", f"{P} pair(s) 8192x8192x128 f32", med, best, 2.0 * n * n * d * P, "FLOP/s",
                   MFMA_F32_PEAK, {"bound": "mfma", "hbm_frac": (2 * n * d * 4 + n * 12) * P / med / HBM_PEAK})
            med, best = timeit(lambda: matching.match_l2(tab, tab, vq, vt, norm_q=nrm, norm_t=nrm, precision=1), args.iters)
            report("l2 filter + re-rank (prep + presplit + l2_filter_kernel<4> + l2_rerank_kernel)", f"{P} pair(s) 8192x8192x128 f32",
                   med, best, 2.0 * n * n * d * P, "FLOP/s", MFMA_F16_PEAK,
                   {"bound": "mfma", "note": "algorithmic FLOP over the nominal f16 peak; 1.03 f16 MFMA FLOP are executed per "
                    "algorithmic FLOP", "over_f32_mfma_peak": 2.0 * n * n * d * P / med / MFMA_F32_PEAK})

    if not only or "hamming" in only:
        for V, n in ((8, 32768), (32, 8192)):
            tab = torch.randint(0, 256, (V, n, 32), dtype=torch.uint8, device=dev)
            i, j = synthetic.all_pairs(V)
            vq, vt = t(i), t(j)
            B = len(i)
            med, best = timeit(lambda: matching.match_hamming(tab, tab, vq, vt), args.iters)
            # the kernel runs 256 fp4 MACs per descriptor pair on the matrix cores (block-scaled fp4 MFMA = 4x the bf16 rate);
            # frac is against that peak, int8_frac against the int8 peak the previous kernel version was priced on
            report("hamming_expand_kernel + hamming_fp4_dma_kernel<2>", f"{B} pairs {n}x{n}x256 bit", med, best,
                   512.0 * n * n * B, "OP/s", MFMA_FP4_PEAK,
                   {"bound": "mfma", "hbm_frac": (2 * n * 32 + n * 12) * B / med / HBM_PEAK, "pairs_per_s": B / med,
                    "int8_frac": 512.0 * n * n * B / med / MFMA_I8_PEAK,
                    "valu_equivalent_frac": 16.0 * n * n * B / med / VALU_PEAK})
        tab = torch.randint(0, 256, (2, 32768, 32), dtype=torch.uint8, device=dev)
        med, best = timeit(lambda: matching.match_hamming(tab[0:1], tab[1:2]), args.iters)
        report("hamming_fp4_dma_kernel+merge", "1 pair 32768x32768x256 bit (train split + merge)", med, best,
               512.0 * 32768 * 32768, "OP/s", MFMA_FP4_PEAK, {"bound": "mfma"})

    if not only or "score_e" in only:
        for B, H, N in ((16, 4096, 8192), (1, 4096, 8192)):
            E = torch.randn((B, H, 9), device=dev)
            pts = [torch.randn((B, N), device=dev) * 0.2 for _ in range(4)]
            med, best = timeit(lambda: ransac.score_essential(E, *pts, 1e-6), args.iters)
            report("score_kernel<Essential>", f"B={B} H={H} N={N}", med, best, 20.0 * B * H * N, "lane-ops/s", VALU_PEAK,
                   {"bound": "valu", "hyps_per_s": B * H / med, "hbm_frac": (16 * N + 40 * H) * B / med / HBM_PEAK})

    if not only or "score_p" in only:
        B, H, N = 32, 16384, 65536
        P = torch.randn((B, H, 12), device=dev)
        pts = [torch.randn((B, N), device=dev) for _ in range(5)]
        med, best = timeit(lambda: ransac.score_pnp(P, *pts, 64.0), max(3, args.iters // 4), warmup=1)
        report("score_kernel<Pnp>", f"B={B} H={H} N={N}", med, best, 17.0 * B * H * N, "lane-ops/s", VALU_PEAK,
               {"bound": "valu", "hyps_per_s": B * H / med, "hbm_frac": (20 * N + 52 * H) * B / med / HBM_PEAK})

    if not only or "tri" in only:
        B, N = 512, 8192
        P1 = t(np.tile(synthetic.projection_matrix(np.eye(3), np.zeros(3), False).astype(np.float32), (B, 1, 1)))
        R = synthetic.rot_y(np.deg2rad(-10.0))
        P2 = t(np.tile(synthetic.projection_matrix(R, -R @ np.array([0.5, 0, 0]), False).astype(np.float32), (B, 1, 1)))
        X = torch.rand((B, N, 3), device=dev) * 2 - 1
        X[..., 2] += 4
        x1, y1 = (X[..., 0] / X[..., 2]).contiguous(), (X[..., 1] / X[..., 2]).contiguous()
        Rt = torch.from_numpy(R).float().to(dev)
        Xc = X @ Rt.T + torch.from_numpy(-R @ np.array([0.5, 0, 0])).float().to(dev)
        x2, y2 = (Xc[..., 0] / Xc[..., 2]).contiguous(), (Xc[..., 1] / Xc[..., 2]).contiguous()
        med, best = timeit(lambda: triangulation.triangulate(P1, P2, x1, y1, x2, y2), args.iters)
        # SURVEY.md §8d prices a point at ~300 FLOP = 150 fma-class lane-ops (4 x 4 DLT system, Jacobi) over 28 bytes; by bytes the
        # kernel would be HBM-bound, in practice the Jacobi sweeps make it VALU-bound, so the VALU fraction is the
        # one reported as `frac`; the byte fraction stays beside it
        report("triangulate_kernel", f"B={B} N={N} C=1", med, best, 150.0 * B * N, "lane-ops/s", VALU_PEAK,
               {"bound": "valu", "points_per_s": B * N / med, "hbm_frac": 28.0 * B * N / med / HBM_PEAK,
                "note": "algorithmic 150 lane-ops per point (SURVEY.md §8d: ~300 FLOP); the kernel executes more (three one-sided Jacobi "
                        "sweeps of a 4 x 4 system in registers)"})


if __name__ == "__main__":
    main()
