"""Brute-force descriptor matching + Lowe ratio test on the GPU.

The reference has no matcher (main.py:146-230 only detects board markers); the semantics are
the contract of include/posehot.h: nearest and second-nearest train row per query row, ties to
the lowest train index, strict '<' ratio test.

Tables are ``[V, n, 32]`` uint8 (binary) or ``[V, n, dim]`` float32; a pair ``b`` matches view
``view_q[b]`` against view ``view_t[b]``.  The C ABI takes at most 65535 pairs per launch; the
wrappers below chunk longer pair lists.
"""
from __future__ import annotations

import torch

from . import _lib as L

MAX_ITEMS = 65535
OWN_NORMS = True    # calls without norm_q / norm_t let the library compute the norms inside the matcher call (False: ph_row_sqnorm
                    # first, the two-pass path — the tests run both)


def _views(table, view, name):
    V = table.shape[0]
    if view is None:
        return V, None
    view = L.dev(view, torch.int32, name)
    return view.numel(), view


def _chunks(B):
    for s in range(0, B, MAX_ITEMS):
        yield s, min(B, s + MAX_ITEMS)


def hamming_view_images(desc_t):
    """The matrix-core images of EVERY view of a ``[V, n, 32]`` uint8 table, expanded once (``ph_hamming_expand_views``; 128 bytes
    per row, resident): hand them to ``match_hamming(..., images=...)`` when the same views are matched again and again."""
    lib = L.load()
    desc_t = L.dev(desc_t, torch.uint8, "desc_t")
    if desc_t.dim() != 3 or desc_t.shape[2] != 32:
        raise ValueError("descriptor table must be [V, n, 32] uint8")
    V, nt = desc_t.shape[0], desc_t.shape[1]
    nbytes = lib.ph_hamming_view_images_bytes(V, nt)
    images = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=desc_t.device)
    L.check(lib.ph_hamming_expand_views(L.ptr(desc_t), V, nt, 32, L.ptr(images), nbytes, L.stream_ptr()), "ph_hamming_expand_views")
    return images


def match_hamming(desc_q, desc_t, view_q=None, view_t=None, n_q=None, n_t=None, images=None):
    """Hamming top-2.  Returns (idx, d1, d2), each int32 ``[B, nq]``.

    images: ``hamming_view_images(desc_t)`` — the train views' images are then read where they lie instead of being expanded
    for the call's pairs (same outputs, bit for bit)."""
    lib = L.load()
    desc_q = L.dev(desc_q, torch.uint8, "desc_q")
    desc_t = L.dev(desc_t, torch.uint8, "desc_t")
    if desc_q.dim() != 3 or desc_t.dim() != 3:
        raise ValueError("descriptor tables must be [V, n, bytes]")
    nq, nt, nbytes = desc_q.shape[1], desc_t.shape[1], desc_q.shape[2]
    if desc_t.shape[2] != nbytes:
        raise ValueError("query and train descriptors differ in width")
    B, view_q = _views(desc_q, view_q, "view_q")
    Bt, view_t = _views(desc_t, view_t, "view_t")
    if B != Bt:
        raise ValueError(f"pair lists differ in length: {B} vs {Bt}")
    n_q, n_t = L.opt_i32(n_q, B, "n_q"), L.opt_i32(n_t, B, "n_t")
    dev = desc_q.device
    idx = torch.empty((B, nq), dtype=torch.int32, device=dev)
    d1 = torch.empty_like(idx)
    d2 = torch.empty_like(idx)
    for s, e in _chunks(B):
        nb = e - s
        sl = lambda t: None if t is None else t[s:e]
        if images is not None:
            if view_t is None and B > 1 and (s, e) != (0, B):
                raise ValueError("resident images with an implicit view list need the whole pair list in one call")
            wsb = lib.ph_match_hamming_images_ws_bytes(nb, nq, nt)
            ws = L.workspace(wsb, dev)
            L.check(lib.ph_match_hamming_top2_images(L.ptr(desc_q if view_q is not None else desc_q[s:e]), L.ptr(desc_t),
                                                     L.ptr(images), L.ptr(sl(view_q)), L.ptr(sl(view_t)), L.ptr(sl(n_q)),
                                                     L.ptr(sl(n_t)), nb, nq, nt, nbytes, L.ptr(idx[s:e]), L.ptr(d1[s:e]),
                                                     L.ptr(d2[s:e]), L.ptr(ws), wsb, L.stream_ptr()), "ph_match_hamming_top2_images")
            continue
        wsb = lib.ph_match_hamming_ws_bytes(nb, nq, nt)
        ws = L.workspace(wsb, dev)
        # with view lists the tables are shared; without, pair b uses view b of each table
        q = desc_q if view_q is not None else desc_q[s:e]
        t = desc_t if view_t is not None else desc_t[s:e]
        L.check(lib.ph_match_hamming_top2(L.ptr(q), L.ptr(t), L.ptr(sl(view_q)), L.ptr(sl(view_t)),
                                          L.ptr(sl(n_q)), L.ptr(sl(n_t)), nb, nq, nt, nbytes,
                                          L.ptr(idx[s:e]), L.ptr(d1[s:e]), L.ptr(d2[s:e]),
                                          L.ptr(ws), wsb, L.stream_ptr()), "ph_match_hamming_top2")
    return idx, d1, d2


def row_sqnorm(table):
    """Squared row norms of a ``[V, n, dim]`` float32 table -> ``[V, n]`` (once per view table)."""
    lib = L.load()
    table = L.dev(table, torch.float32, "table")
    out = torch.empty(table.shape[:-1], dtype=torch.float32, device=table.device)
    L.check(lib.ph_row_sqnorm(L.ptr(table), out.numel(), table.shape[-1], L.ptr(out), L.stream_ptr()), "ph_row_sqnorm")
    return out


def match_l2(desc_q, desc_t, view_q=None, view_t=None, n_q=None, n_t=None, precision=L.PRECISION_F16_RERANK,
             norm_q=None, norm_t=None, kernel_events=None):
    """Squared-L2 top-2.  Returns (idx int32, d1sq f32, d2sq f32), each ``[B, nq]``.

    precision picks the algorithm, not the result (both follow the f32 fma-chain contract bit for bit):
    ``L.PRECISION_F32`` scores every pair of rows on the f32 MFMA; ``L.PRECISION_F16_RERANK`` (default) filters
    candidates with one fp16 MFMA product per score (inside a proven error band) and re-ranks them with the exact chain.

    norm_q / norm_t: cached ``row_sqnorm`` of the tables — views that many pairs share (cfg 3 / 4).  When neither is given the
    fp16-filter path computes the norms inside the call, fused into the passes that read the two tables anyway (one read of
    each table instead of two); the outputs are the same bit for bit.
    kernel_events: optional pair of raw hipEvent_t handles (ints) recorded around the call's dominant kernel alone
    (``ph_match_l2_top2_timed``; measurement only, at most one launch chunk).
    """
    lib = L.load()
    desc_q = L.dev(desc_q, torch.float32, "desc_q")
    desc_t = L.dev(desc_t, torch.float32, "desc_t")
    if desc_q.dim() != 3 or desc_t.dim() != 3:
        raise ValueError("descriptor tables must be [V, n, dim]")
    nq, nt, dim = desc_q.shape[1], desc_t.shape[1], desc_q.shape[2]
    if desc_t.shape[2] != dim:
        raise ValueError("query and train descriptors differ in width")
    B, view_q = _views(desc_q, view_q, "view_q")
    Bt, view_t = _views(desc_t, view_t, "view_t")
    if B != Bt:
        raise ValueError(f"pair lists differ in length: {B} vs {Bt}")
    n_q, n_t = L.opt_i32(n_q, B, "n_q"), L.opt_i32(n_t, B, "n_t")
    dev = desc_q.device
    # no norms given: the fp16-filter path computes them inside the call, in the passes that read the tables anyway (NULL to
    # the library); the f32 kernel needs them up front
    own_norms = OWN_NORMS and norm_q is None and norm_t is None and int(precision) == L.PRECISION_F16_RERANK and dim == 128
    if not own_norms:
        norm_q = row_sqnorm(desc_q) if norm_q is None else L.dev(norm_q, torch.float32, "norm_q")
        norm_t = (norm_q if desc_t is desc_q else row_sqnorm(desc_t)) if norm_t is None else L.dev(norm_t, torch.float32, "norm_t")
    idx = torch.empty((B, nq), dtype=torch.int32, device=dev)
    d1 = torch.empty((B, nq), dtype=torch.float32, device=dev)
    d2 = torch.empty_like(d1)
    for s, e in _chunks(B):
        nb = e - s
        wsb = lib.ph_match_l2_ws_bytes(nb, nq, nt, dim, int(precision))
        ws = L.workspace(wsb, dev)
        sl = lambda t: None if t is None else t[s:e]
        q = desc_q if view_q is not None else desc_q[s:e]
        t = desc_t if view_t is not None else desc_t[s:e]
        qn = None if own_norms else (norm_q if view_q is not None else norm_q[s:e])
        tn = None if own_norms else (norm_t if view_t is not None else norm_t[s:e])
        args = (L.ptr(q), L.ptr(t), L.ptr(qn), L.ptr(tn), L.ptr(sl(view_q)), L.ptr(sl(view_t)),
                L.ptr(sl(n_q)), L.ptr(sl(n_t)), nb, nq, nt, dim, precision,
                L.ptr(idx[s:e]), L.ptr(d1[s:e]), L.ptr(d2[s:e]), L.ptr(ws), wsb, L.stream_ptr())
        if kernel_events is not None:
            L.check(lib.ph_match_l2_top2_timed(*args, kernel_events[0], kernel_events[1]), "ph_match_l2_top2_timed")
        else:
            L.check(lib.ph_match_l2_top2(*args), "ph_match_l2_top2")
    return idx, d1, d2


def ratio_filter_hamming(idx, d1, d2, ratio=0.75):
    """keep = second exists and d1 < ratio*d2 (strict).  int32 mask shaped like idx."""
    lib = L.load()
    idx, d1, d2 = L.dev(idx, torch.int32, "idx"), L.dev(d1, torch.int32, "d1"), L.dev(d2, torch.int32, "d2")
    keep = torch.empty_like(idx)
    L.check(lib.ph_ratio_filter_hamming(L.ptr(idx), L.ptr(d1), L.ptr(d2), idx.numel(), float(ratio),
                                        L.ptr(keep), L.stream_ptr()), "ph_ratio_filter_hamming")
    return keep


def match_l2_ratio(desc_q, desc_t, ratio, view_q=None, view_t=None, n_q=None, n_t=None, norm_q=None, norm_t=None,
                   kernel_events=None):
    """Squared-L2 matching and Lowe's ratio test in one call (``ph_match_l2_ratio``).  Returns (idx int32, keep int32), each
    ``[B, nq]``: keep is bit for bit what ``match_l2`` (default precision) followed by ``ratio_filter_l2`` return, idx their
    idx where keep is set and -1 elsewhere; no distances.  A query whose decision already follows from the candidate
    filter's approximate scores is never re-scored exactly.  dim = 128 only."""
    lib = L.load()
    desc_q = L.dev(desc_q, torch.float32, "desc_q")
    desc_t = L.dev(desc_t, torch.float32, "desc_t")
    if desc_q.dim() != 3 or desc_t.dim() != 3:
        raise ValueError("descriptor tables must be [V, n, dim]")
    nq, nt, dim = desc_q.shape[1], desc_t.shape[1], desc_q.shape[2]
    if desc_t.shape[2] != dim:
        raise ValueError("query and train descriptors differ in width")
    B, view_q = _views(desc_q, view_q, "view_q")
    Bt, view_t = _views(desc_t, view_t, "view_t")
    if B != Bt:
        raise ValueError(f"pair lists differ in length: {B} vs {Bt}")
    n_q, n_t = L.opt_i32(n_q, B, "n_q"), L.opt_i32(n_t, B, "n_t")
    dev = desc_q.device
    own_norms = OWN_NORMS and norm_q is None and norm_t is None        # computed inside the call (NULL to the library), one pass per table
    if not own_norms:
        norm_q = row_sqnorm(desc_q) if norm_q is None else L.dev(norm_q, torch.float32, "norm_q")
        norm_t = (norm_q if desc_t is desc_q else row_sqnorm(desc_t)) if norm_t is None else L.dev(norm_t, torch.float32, "norm_t")
    idx = torch.empty((B, nq), dtype=torch.int32, device=dev)
    keep = torch.empty((B, nq), dtype=torch.int32, device=dev)
    ev = kernel_events if kernel_events is not None else (None, None)
    for s, e in _chunks(B):
        nb = e - s
        wsb = lib.ph_match_l2_ws_bytes(nb, nq, nt, dim, int(L.PRECISION_F16_RERANK))
        ws = L.workspace(wsb, dev)
        sl = lambda t: None if t is None else t[s:e]
        q = desc_q if view_q is not None else desc_q[s:e]
        t = desc_t if view_t is not None else desc_t[s:e]
        qn = None if own_norms else (norm_q if view_q is not None else norm_q[s:e])
        tn = None if own_norms else (norm_t if view_t is not None else norm_t[s:e])
        L.check(lib.ph_match_l2_ratio(L.ptr(q), L.ptr(t), L.ptr(qn), L.ptr(tn), L.ptr(sl(view_q)), L.ptr(sl(view_t)),
                                      L.ptr(sl(n_q)), L.ptr(sl(n_t)), nb, nq, nt, dim, float(ratio),
                                      L.ptr(idx[s:e]), L.ptr(keep[s:e]), L.ptr(ws), wsb, L.stream_ptr(), ev[0], ev[1]),
                "ph_match_l2_ratio")
    return idx, keep


def ratio_filter_l2(idx, d1sq, d2sq, ratio=0.75):
    """keep = second exists and d1sq < ratio^2*d2sq (strict; == ratio test on true distances)."""
    lib = L.load()
    idx = L.dev(idx, torch.int32, "idx")
    d1sq, d2sq = L.dev(d1sq, torch.float32, "d1sq"), L.dev(d2sq, torch.float32, "d2sq")
    keep = torch.empty_like(idx)
    L.check(lib.ph_ratio_filter_l2(L.ptr(idx), L.ptr(d1sq), L.ptr(d2sq), idx.numel(), float(ratio),
                                   L.ptr(keep), L.stream_ptr()), "ph_ratio_filter_l2")
    return keep


def compact_matches(idx, keep, kp_q, kp_t, view_q=None, view_t=None, want_indices=True):
    """Stable compaction of kept matches into SoA correspondences.

    kp_q/kp_t: ``[V, n, 2]`` float32 keypoints.  Returns a dict with x1,y1,x2,y2 ``[B, nq]``
    (first n_match[b] entries valid), n_match ``[B]`` and optionally match_q/match_t.
    """
    lib = L.load()
    idx, keep = L.dev(idx, torch.int32, "idx"), L.dev(keep, torch.int32, "keep")
    kp_q, kp_t = L.dev(kp_q, torch.float32, "kp_q"), L.dev(kp_t, torch.float32, "kp_t")
    B, nq = idx.shape
    nt = kp_t.shape[1]
    if kp_q.shape[1] != nq or kp_q.shape[2] != 2 or kp_t.shape[2] != 2:
        raise ValueError("keypoint tables must be [V, n, 2] with n matching the descriptor tables")
    view_q = None if view_q is None else L.dev(view_q, torch.int32, "view_q")
    view_t = None if view_t is None else L.dev(view_t, torch.int32, "view_t")
    dev = idx.device
    out = {k: torch.empty((B, nq), dtype=torch.float32, device=dev) for k in ("x1", "y1", "x2", "y2")}
    out["n_match"] = torch.empty((B,), dtype=torch.int32, device=dev)
    mq = mt = None
    if want_indices:
        mq = torch.empty((B, nq), dtype=torch.int32, device=dev)
        mt = torch.empty_like(mq)
        out["match_q"], out["match_t"] = mq, mt
    L.check(lib.ph_compact_matches(L.ptr(idx), L.ptr(keep), L.ptr(kp_q), L.ptr(kp_t), L.ptr(view_q), L.ptr(view_t),
                                   B, nq, nt, L.ptr(out["x1"]), L.ptr(out["y1"]), L.ptr(out["x2"]), L.ptr(out["y2"]),
                                   L.ptr(mq), L.ptr(mt), L.ptr(out["n_match"]), L.stream_ptr()), "ph_compact_matches")
    return out
