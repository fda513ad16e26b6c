"""Batched linear (DLT) triangulation + cheirality vote on the GPU.

No reference code.  ``P1``/``P2`` are ``[B, C, 12]`` float32 candidate projection pairs
(C = 1: plain triangulation; C = 4: the four (R, t) candidates of an essential matrix).
"""
from __future__ import annotations

import torch

from . import _lib as L


def triangulate(P1, P2, x1, y1, x2, y2, n_valid=None, mask=None, want_points=True, want_front=False):
    """Returns (XYZ ``[B, C, N, 3]`` or None, front ``[B, C]`` int32 or None)."""
    lib = L.load()
    P1, P2 = L.dev(P1, torch.float32, "P1"), L.dev(P2, torch.float32, "P2")
    if P1.dim() != 3 or P1.shape[2] != 12 or P2.shape != P1.shape:
        raise ValueError("P1/P2 must both be [B, C, 12]")
    B, Cn = P1.shape[0], P1.shape[1]
    pts = [L.dev(a, torch.float32, n) for a, n in zip((x1, y1, x2, y2), ("x1", "y1", "x2", "y2"))]
    N = pts[0].shape[1]
    if any(a.shape != (B, N) for a in pts):
        raise ValueError("point arrays must be [B, N]")
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    if mask is not None:
        mask = L.dev(mask, torch.int64, "mask")
        if mask.shape != (B, (N + 63) // 64):
            raise ValueError("mask must be [B, ceil(N/64)] 64-bit words")
    dev = P1.device
    XYZ = torch.empty((B, Cn, N, 3), dtype=torch.float32, device=dev) if want_points else None
    front = torch.empty((B, Cn), dtype=torch.int32, device=dev) if want_front else None
    if XYZ is None and front is None:
        raise ValueError("nothing requested")
    L.check(lib.ph_triangulate(L.ptr(P1), L.ptr(P2), *(L.ptr(a) for a in pts), L.ptr(n_valid), L.ptr(mask),
                               B, Cn, N, L.ptr(XYZ), L.ptr(front), L.stream_ptr()), "ph_triangulate")
    return XYZ, front


def pose_table(P2, front, best_count, n_match, best_idx, min_inliers=5):
    """Cheirality winner + pose-table rows.  P2 ``[B, C, 12]`` f32, front int32 ``[B, C]``; returns
    (table ``[B, 16]`` f32, win int32 ``[B]``)."""
    lib = L.load()
    P2 = L.dev(P2, torch.float32, "P2")
    B, C = P2.shape[0], P2.shape[1]
    front = L.dev(front, torch.int32, "front")
    best_count, n_match, best_idx = (L.opt_i32(t, B, n) for t, n in ((best_count, "best_count"), (n_match, "n_match"),
                                                                      (best_idx, "best_idx")))
    table = torch.empty((B, 16), dtype=torch.float32, device=P2.device)
    win = torch.empty((B,), dtype=torch.int32, device=P2.device)
    L.check(lib.ph_pose_table(L.ptr(P2), L.ptr(front), C, L.ptr(best_count), L.ptr(n_match), L.ptr(best_idx), B,
                              int(min_inliers), L.ptr(table), L.ptr(win), L.stream_ptr()), "ph_pose_table")
    return table, win
