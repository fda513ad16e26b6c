"""RANSAC hypothesis scoring on the GPU.

PnP residual = the reference's reprojection error (validate_pose.py:260-288) on undistorted
points; the essential-matrix (Sampson) side has no reference code.  Inputs are SoA point rows
``[B, N]`` float32 and models ``[B, H, 9]`` (E, row-major) or ``[B, H, 12]`` (3x4 projection).
"""
from __future__ import annotations

import torch

from . import _lib as L


def _pts(arrs, names):
    out = [L.dev(a, torch.float32, n) for a, n in zip(arrs, names)]
    shp = out[0].shape
    if len(shp) != 2 or any(a.shape != shp for a in out):
        raise ValueError("point arrays must share one [B, N] shape")
    return out


def _models(m, B, width, name):
    m = L.dev(m, torch.float32, name)
    if m.dim() != 3 or m.shape[0] != B or m.shape[2] != width:
        raise ValueError(f"{name}: expected [B={B}, H, {width}], got {tuple(m.shape)}")
    return m


def sample_minimal(seed, n_valid, N, n_samples, m=5, item_key=None):
    """Counter-based (Philox4x32-10) minimal samples: int32 ``[B, n_samples, m]`` distinct indices per row.

    item_key: int32/int64 ``[B]`` device tensor of per-item Philox keys (e.g. global pair ids, taken mod 2^32) or None
    (= the item's position in the batch): with keys, an item draws the same samples wherever it sits in a batch."""
    lib = L.load()
    n_valid = L.dev(n_valid, torch.int32, "n_valid")
    B = n_valid.numel()
    out = torch.empty((B, n_samples, m), dtype=torch.int32, device=n_valid.device)
    if item_key is not None:
        if not isinstance(item_key, torch.Tensor) or not item_key.is_cuda or item_key.numel() != B:
            raise ValueError("item_key must be a device tensor with one key per item")
        item_key = (item_key.to(torch.int64) & 0xFFFFFFFF).to(torch.int32) if item_key.dtype != torch.int32 else item_key
        item_key = item_key.contiguous()
    L.check(lib.ph_sample_minimal_keyed(int(seed) & 0xFFFFFFFF, L.ptr(item_key), L.ptr(n_valid), B, int(N), int(n_samples),
                                        int(m), L.ptr(out), L.stream_ptr()), "ph_sample_minimal_keyed")
    return out


NOALLOC_MAX_SAMPLES = 8192   # PH_NOALLOC_MAX_SAMPLES (posehot.h)


def solve_essential_5pt(x1, y1, x2, y2, samples, want_counts=False, noalloc=False):
    """Nister 5-point solver on the device.  samples int32 ``[B, S, 5]`` -> E ``[B, 10*S, 9]`` float32
    (10 slots per sample, slot = root index, non-real roots give all-zero slots) and optionally the per-sample solution counts ``[B, S]``.

    noalloc=True: ``ph_solve_essential_5pt_noalloc`` — the polynomial records live in the library's static device scratch
    instead of a workspace tensor (B * S <= 8192; calls overlapping on one GPU race on it); same outputs, bit for bit."""
    lib = L.load()
    x1, y1, x2, y2 = _pts((x1, y1, x2, y2), ("x1", "y1", "x2", "y2"))
    B, N = x1.shape
    samples = L.dev(samples, torch.int32, "samples")
    if samples.dim() != 3 or samples.shape[0] != B or samples.shape[2] != 5:
        raise ValueError("samples must be [B, S, 5]")
    S = samples.shape[1]
    E = torch.empty((B, 10 * S, 9), dtype=torch.float32, device=x1.device)
    ns = torch.empty((B, S), dtype=torch.int32, device=x1.device) if want_counts else None
    if noalloc:
        L.check(lib.ph_solve_essential_5pt_noalloc(L.ptr(x1), L.ptr(y1), L.ptr(x2), L.ptr(y2), L.ptr(samples), B, S, N,
                                                   L.ptr(E), L.ptr(ns), L.stream_ptr()), "ph_solve_essential_5pt_noalloc")
        return (E, ns) if want_counts else E
    wsb = lib.ph_solve_essential_5pt_ws_bytes(B, S)
    ws = L.workspace(wsb, x1.device)
    L.check(lib.ph_solve_essential_5pt(L.ptr(x1), L.ptr(y1), L.ptr(x2), L.ptr(y2), L.ptr(samples), B, S, N, L.ptr(E),
                                       L.ptr(ns), L.ptr(ws), wsb, L.stream_ptr()), "ph_solve_essential_5pt")
    return (E, ns) if want_counts else E


def solve_p3p(X, Y, Z, u, v, samples, K4=None, want_counts=False):
    """P3P solver on the device.  samples int32 ``[B, S, 3]``; (u, v) normalised undistorted coordinates ->
    P ``[B, 4*S, 12]`` float32 (4 slots per sample, slot = root index, empty slots all-zero): ``K [R|t]`` when
    K4 = (fx, fy, cx, cy) (float64 device tensor) is given, else ``[R|t]``.  Optionally the solution counts ``[B, S]``."""
    lib = L.load()
    X, Y, Z, u, v = _pts((X, Y, Z, u, v), ("X", "Y", "Z", "u", "v"))
    B, N = X.shape
    samples = L.dev(samples, torch.int32, "samples")
    if samples.dim() != 3 or samples.shape[0] != B or samples.shape[2] != 3:
        raise ValueError("samples must be [B, S, 3]")
    if K4 is not None:
        K4 = L.dev(K4, torch.float64, "K4")
    S = samples.shape[1]
    P = torch.empty((B, 4 * S, 12), dtype=torch.float32, device=X.device)
    ns = torch.empty((B, S), dtype=torch.int32, device=X.device) if want_counts else None
    L.check(lib.ph_solve_p3p(L.ptr(X), L.ptr(Y), L.ptr(Z), L.ptr(u), L.ptr(v), L.ptr(samples), B, S, N, L.ptr(K4),
                             L.ptr(P), L.ptr(ns), L.stream_ptr()), "ph_solve_p3p")
    return (P, ns) if want_counts else P


def gather_models(models, sel):
    """models ``[B, H, stride]`` float32, sel int32 ``[B]`` -> ``[B, stride]``: hypothesis sel[b] of every item."""
    lib = L.load()
    models = L.dev(models, torch.float32, "models")
    B, H, stride = models.shape
    sel = L.opt_i32(sel, B, "sel")
    out = torch.empty((B, stride), dtype=torch.float32, device=models.device)
    L.check(lib.ph_gather_models(L.ptr(models), L.ptr(sel), B, H, stride, L.ptr(out), L.stream_ptr()), "ph_gather_models")
    return out


def compact_models(models, counts=None):
    """Live (non-zero) models of every item first, in slot order: returns (compact ``[B, H, stride]`` with a zero tail,
    slot map int32 ``[B, H]`` (-1 in the tail), n_live int32 ``[B]``).

    counts: int32 ``[B, H // group]`` — the solvers' ``n_solutions`` (live models per minimal sample's group of slots); with
    them an item is packed by many blocks instead of one (``ph_compact_models_counted``).  They must be the true counts."""
    lib = L.load()
    models = L.dev(models, torch.float32, "models")
    if models.dim() != 3:
        raise ValueError("models must be [B, H, stride]")
    B, H, stride = models.shape
    out = torch.empty_like(models)
    slot = torch.empty((B, H), dtype=torch.int32, device=models.device)
    n_live = torch.empty((B,), dtype=torch.int32, device=models.device)
    if counts is not None:
        counts = L.dev(counts, torch.int32, "counts")
        if counts.dim() != 2 or counts.shape[0] != B or counts.shape[1] < 1 or H % counts.shape[1] != 0:
            raise ValueError("counts must be [B, H // group]")
        L.check(lib.ph_compact_models_counted(L.ptr(models), L.ptr(counts), H // counts.shape[1], B, H, stride, L.ptr(out),
                                              L.ptr(slot), L.ptr(n_live), L.stream_ptr()), "ph_compact_models_counted")
        return out, slot, n_live
    L.check(lib.ph_compact_models(L.ptr(models), B, H, stride, L.ptr(out), L.ptr(slot), L.ptr(n_live), L.stream_ptr()),
            "ph_compact_models")
    return out, slot, n_live


def lookup_index(table, sel):
    """``table[b, sel[b]]`` int32 ``[B]`` (-1 where sel[b] is out of range): a compact winner's slot."""
    lib = L.load()
    table = L.dev(table, torch.int32, "table")
    B, H = table.shape
    sel = L.opt_i32(sel, B, "sel")
    out = torch.empty((B,), dtype=torch.int32, device=table.device)
    L.check(lib.ph_lookup_index(L.ptr(table), L.ptr(sel), B, H, L.ptr(out), L.stream_ptr()), "ph_lookup_index")
    return out


def decompose_essential(E, sel=None):
    """The four (R, t) candidates of hypothesis sel[b] of each item's ``[H, 9]`` block, on the device.

    Returns (P1, P2) float32 ``[B, 4, 12]``: P1 = [I | 0], P2 = [R | t] in the order (Ra, t), (Ra, -t), (Rb, t),
    (Rb, -t).  Device counterpart of ``geometry.candidate_projections`` (same candidate set, own order)."""
    lib = L.load()
    E = L.dev(E, torch.float32, "E")
    if E.dim() != 3 or E.shape[2] != 9:
        raise ValueError("E must be [B, H, 9]")
    B, H = E.shape[0], E.shape[1]
    sel = L.opt_i32(sel, B, "sel")
    P1 = torch.empty((B, 4, 12), dtype=torch.float32, device=E.device)
    P2 = torch.empty((B, 4, 12), dtype=torch.float32, device=E.device)
    L.check(lib.ph_decompose_essential(L.ptr(E), L.ptr(sel), B, H, L.ptr(P1), L.ptr(P2), L.stream_ptr()),
            "ph_decompose_essential")
    return P1, P2


def refine_relative_pose(x1, y1, x2, y2, pose_rt, trunc_sq, n_valid=None, mask=None, max_iters=4, want_stats=False):
    """LM refinement of B relative poses IN PLACE on the truncated Sampson cost (``ph_refine_relative_pose``).

    pose_rt: float32 ``[B, stride >= 12]`` rows of R (9, row-major) | t (3) — e.g. the ``[B, 16]`` pose table.
    trunc_sq: squared truncation distance in the points' units (the pipeline uses 9 x the RANSAC thr_sq).
    Returns pose_rt, or (pose_rt, rms f64 [B], n_inliers i32 [B], iters i32 [B]) with want_stats."""
    lib = L.load()
    x1, y1, x2, y2 = _pts((x1, y1, x2, y2), ("x1", "y1", "x2", "y2"))
    B, N = x1.shape
    # the in-place operand is checked on the CALLER's tensor (L.dev would hand back a contiguous copy of a strided one, and
    # the refinement would silently land in the copy)
    if not isinstance(pose_rt, torch.Tensor) or not pose_rt.is_cuda or pose_rt.dtype != torch.float32:
        raise L.PosehotError("pose_rt must be a float32 tensor on the GPU (there is no CPU path)")
    if pose_rt.dim() != 2 or pose_rt.shape[0] != B or pose_rt.shape[1] < 12:
        raise ValueError("pose_rt must be [B, >= 12] float32")
    if not pose_rt.is_contiguous():
        raise ValueError("pose_rt is refined in place and must be contiguous")
    if mask is not None:
        mask = L.dev(mask, torch.int64, "mask")
        if mask.shape != (B, (N + 63) // 64):
            raise ValueError("mask must be [B, ceil(N/64)] 64-bit words")
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    rms = nin = iters = None
    if want_stats:
        rms = torch.empty((B,), dtype=torch.float64, device=x1.device)
        nin = torch.empty((B,), dtype=torch.int32, device=x1.device)
        iters = torch.empty((B,), dtype=torch.int32, device=x1.device)
    L.check(lib.ph_refine_relative_pose(L.ptr(x1), L.ptr(y1), L.ptr(x2), L.ptr(y2), L.ptr(mask), L.ptr(n_valid), B, N,
                                        float(trunc_sq), int(max_iters), L.ptr(pose_rt), pose_rt.shape[1], L.ptr(rms),
                                        L.ptr(nin), L.ptr(iters), L.stream_ptr()), "ph_refine_relative_pose")
    return (pose_rt, rms, nin, iters) if want_stats else pose_rt


def pose_from_projection(P, sel=None, K4=None):
    """(rvec, tvec) float64 ``[B, 6]`` of hypothesis sel[b] of each item's ``[H, 12]`` block (K [R|t] when K4 is given)."""
    lib = L.load()
    P = L.dev(P, torch.float32, "P")
    if P.dim() != 3 or P.shape[2] != 12:
        raise ValueError("P must be [B, H, 12]")
    B, H = P.shape[0], P.shape[1]
    sel = L.opt_i32(sel, B, "sel")
    if K4 is not None:
        K4 = L.dev(K4, torch.float64, "K4")
    pose = torch.empty((B, 6), dtype=torch.float64, device=P.device)
    L.check(lib.ph_pose_from_projection(L.ptr(P), L.ptr(sel), H, B, L.ptr(K4), L.ptr(pose), L.stream_ptr()),
            "ph_pose_from_projection")
    return pose


def refine_pose_lm_soa(X, Y, Z, u, v, pose, K4, dist5=None, mask=None, n_valid=None, max_iters=30, blocks="auto"):
    """Levenberg-Marquardt on the reprojection error over the masked points (float64).
    Returns (pose ``[B, 6]``, rms ``[B]``, iters ``[B]``); the input pose tensor is not modified.

    blocks="auto": large point sets in small batches run several blocks per item, one launch per iteration
    (``ph_refine_pose_lm_soa_ws``); "one": always the one-block-per-item kernel (``ph_refine_pose_lm_soa``).  The two agree
    to rounding."""
    if blocks not in ("auto", "one"):
        raise ValueError("blocks must be 'auto' or 'one'")
    lib = L.load()
    X, Y, Z, u, v = _pts((X, Y, Z, u, v), ("X", "Y", "Z", "u", "v"))
    B, N = X.shape
    pose = L.dev(pose, torch.float64, "pose").clone()
    K4 = L.dev(K4, torch.float64, "K4")
    dist5 = torch.zeros(5, dtype=torch.float64, device=X.device) if dist5 is None else L.dev(dist5, torch.float64, "dist5")
    if mask is not None:
        mask = L.dev(mask, torch.int64, "mask")
        if mask.shape != (B, (N + 63) // 64):
            raise ValueError("mask must be [B, ceil(N/64)] 64-bit words")
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    rms = torch.empty((B,), dtype=torch.float64, device=X.device)
    iters = torch.empty((B,), dtype=torch.int32, device=X.device)
    nws = int(lib.ph_refine_pose_lm_soa_ws_bytes(B, N)) if blocks == "auto" else 0
    if nws > 0:
        ws = L.workspace(nws, X.device)
        L.check(lib.ph_refine_pose_lm_soa_ws(L.ptr(X), L.ptr(Y), L.ptr(Z), L.ptr(u), L.ptr(v), L.ptr(mask), L.ptr(n_valid), B, N,
                                             L.ptr(K4), L.ptr(dist5), int(max_iters), L.ptr(pose), L.ptr(rms), L.ptr(iters),
                                             L.ptr(ws), nws, L.stream_ptr()), "ph_refine_pose_lm_soa_ws")
        return pose, rms, iters
    L.check(lib.ph_refine_pose_lm_soa(L.ptr(X), L.ptr(Y), L.ptr(Z), L.ptr(u), L.ptr(v), L.ptr(mask), L.ptr(n_valid), B, N,
                                      L.ptr(K4), L.ptr(dist5), int(max_iters), L.ptr(pose), L.ptr(rms), L.ptr(iters),
                                      L.stream_ptr()), "ph_refine_pose_lm_soa")
    return pose, rms, iters


def score_essential(E, x1, y1, x2, y2, thr_sq, n_valid=None, bailout=False):
    """Inlier count of every E hypothesis (Sampson error < thr).  Returns int32 ``[B, H]``.

    bailout=True: hypotheses that can no longer reach the item's best count are abandoned; the counts of the
    maximum (what ``select_best`` reads) stay exact, the others become lower bounds."""
    lib = L.load()
    x1, y1, x2, y2 = _pts((x1, y1, x2, y2), ("x1", "y1", "x2", "y2"))
    B, N = x1.shape
    E = _models(E, B, 9, "E")
    H = E.shape[1]
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    count = torch.empty((B, H), dtype=torch.int32, device=E.device)
    if bailout:
        lb = torch.empty((B, L.BAILOUT_SLOTS), dtype=torch.int32, device=E.device)
        L.check(lib.ph_score_essential_bailout(L.ptr(E), L.ptr(x1), L.ptr(y1), L.ptr(x2), L.ptr(y2), L.ptr(n_valid),
                                               B, H, N, float(thr_sq), L.ptr(count), L.ptr(lb), L.stream_ptr()),
                "ph_score_essential_bailout")
        return count
    L.check(lib.ph_score_essential(L.ptr(E), L.ptr(x1), L.ptr(y1), L.ptr(x2), L.ptr(y2), L.ptr(n_valid),
                                   B, H, N, float(thr_sq), L.ptr(count), L.stream_ptr()), "ph_score_essential")
    return count


def score_pnp(P, X, Y, Z, u, v, thr_sq, n_valid=None, bailout=False):
    """Inlier count of every 3x4 projection hypothesis.  Returns int32 ``[B, H]`` (bailout: see ``score_essential``)."""
    lib = L.load()
    X, Y, Z, u, v = _pts((X, Y, Z, u, v), ("X", "Y", "Z", "u", "v"))
    B, N = X.shape
    P = _models(P, B, 12, "P")
    H = P.shape[1]
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    count = torch.empty((B, H), dtype=torch.int32, device=P.device)
    if bailout:
        lb = torch.empty((B, L.BAILOUT_SLOTS), dtype=torch.int32, device=P.device)
        L.check(lib.ph_score_pnp_bailout(L.ptr(P), L.ptr(X), L.ptr(Y), L.ptr(Z), L.ptr(u), L.ptr(v), L.ptr(n_valid),
                                         B, H, N, float(thr_sq), L.ptr(count), L.ptr(lb), L.stream_ptr()),
                "ph_score_pnp_bailout")
        return count
    L.check(lib.ph_score_pnp(L.ptr(P), L.ptr(X), L.ptr(Y), L.ptr(Z), L.ptr(u), L.ptr(v), L.ptr(n_valid),
                             B, H, N, float(thr_sq), L.ptr(count), L.stream_ptr()), "ph_score_pnp")
    return count


def select_best(count, slot_map=None):
    """argmax over hypotheses, ties to the lowest index.  Returns (best_idx, best_count) int32 ``[B]``.

    slot_map: the slot map of ``compact_models`` when `count` are the counts of its compact block — best_idx is then the
    winner's SLOT (``ph_select_best_mapped``: arg-max and ``lookup_index`` in one launch; -1 for an item without a live model)."""
    lib = L.load()
    count = L.dev(count, torch.int32, "count")
    B, H = count.shape
    bi = torch.empty((B,), dtype=torch.int32, device=count.device)
    bc = torch.empty_like(bi)
    if slot_map is not None:
        slot_map = L.dev(slot_map, torch.int32, "slot_map")
        if slot_map.shape != (B, H):
            raise ValueError("slot_map must be [B, H] like count")
        L.check(lib.ph_select_best_mapped(L.ptr(count), L.ptr(slot_map), B, H, L.ptr(bi), L.ptr(bc), L.stream_ptr()),
                "ph_select_best_mapped")
        return bi, bc
    L.check(lib.ph_select_best(L.ptr(count), B, H, L.ptr(bi), L.ptr(bc), L.stream_ptr()), "ph_select_best")
    return bi, bc


def inlier_mask_essential(E, sel, x1, y1, x2, y2, thr_sq, n_valid=None):
    """64-bit inlier words ``[B, ceil(N/64)]`` (int64 tensor holding the bit patterns) of model sel[b]."""
    lib = L.load()
    x1, y1, x2, y2 = _pts((x1, y1, x2, y2), ("x1", "y1", "x2", "y2"))
    B, N = x1.shape
    E = _models(E, B, 9, "E")
    sel = L.opt_i32(sel, B, "sel")
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    mask = torch.empty((B, (N + 63) // 64), dtype=torch.int64, device=E.device)      # the kernel writes every word
    L.check(lib.ph_inlier_mask_essential(L.ptr(E), L.ptr(sel), E.shape[1], L.ptr(x1), L.ptr(y1), L.ptr(x2), L.ptr(y2),
                                         L.ptr(n_valid), B, N, float(thr_sq), L.ptr(mask), L.stream_ptr()),
            "ph_inlier_mask_essential")
    return mask


def inlier_mask_pnp(P, sel, X, Y, Z, u, v, thr_sq, n_valid=None):
    lib = L.load()
    X, Y, Z, u, v = _pts((X, Y, Z, u, v), ("X", "Y", "Z", "u", "v"))
    B, N = X.shape
    P = _models(P, B, 12, "P")
    sel = L.opt_i32(sel, B, "sel")
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    mask = torch.empty((B, (N + 63) // 64), dtype=torch.int64, device=P.device)      # the kernel writes every word
    L.check(lib.ph_inlier_mask_pnp(L.ptr(P), L.ptr(sel), P.shape[1], L.ptr(X), L.ptr(Y), L.ptr(Z), L.ptr(u), L.ptr(v),
                                   L.ptr(n_valid), B, N, float(thr_sq), L.ptr(mask), L.stream_ptr()),
            "ph_inlier_mask_pnp")
    return mask


def unpack_mask(mask, N):
    """[B, words] int64 bit words -> bool [B, N] (host-side convenience for tests and reports)."""
    bits = torch.arange(64, device=mask.device, dtype=torch.int64)
    return ((mask.unsqueeze(-1) >> bits) & 1).bool().reshape(mask.shape[0], -1)[:, :N]
