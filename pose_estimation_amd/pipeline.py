"""Batched two-view relative-pose pipeline: the hot path north_star describes, end to end.

    match (L2 or Hamming, top-2) -> Lowe ratio test -> compact correspondences
      -> Philox minimal samples -> 5-point solver (hypotheses) -> score H hypotheses
      -> best hypothesis -> inlier mask                                         [front, device]
      -> decompose the winning E into its 4 (R, t) candidates                   [device, or host 3x3 SVDs]
      -> triangulate with all 4 + cheirality vote -> pose table row             [back, device]

The reference's per-image loop (main.py:163-214) is what gets batched: one launch per stage
handles B pairs.  north_star leaves the recoverPose-style decomposition on the host; with
``decompose="host"`` the pipeline does exactly that (3x3 SVD per pair, one host round trip) and
is split at the round trip into `relative_pose_front` / `relative_pose_back`, so a caller can
enqueue the next batch's front before finishing the previous batch's back.  The default
``decompose="device"`` uses the closed-form device decomposition (`ph_decompose_essential`)
instead: the whole batch is then one chain of launches with no host synchronisation at all.
Output: a [B, 16] float32 pose table (R row-major 9, t 3, inliers, matches, winning hypothesis,
status) — the unit that is gathered across ranks.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import geometry, matching, ransac, triangulation

REFINE_TRUNC = 3.0          # truncation of the refinement's loss, in RANSAC thresholds
REFINE_ITERS = 4            # LM iterations (one pass over the correspondences each): converged to < 0.001 deg by then
STATUS_OK = 1.0
STATUS_DEGENERATE = 0.0



@dataclass
class Front:
    """Device results of the front half plus the in-flight host copy of the winning E matrices."""
    pts: tuple
    n_match: torch.Tensor
    E_hyp: torch.Tensor
    best_idx: torch.Tensor
    best_count: torch.Tensor
    mask: torch.Tensor
    E_best_host: torch.Tensor            # None with decompose="device"
    copied: torch.cuda.Event             # None with decompose="device"
    want_points: bool = False
    extra: dict = field(default_factory=dict)
    opts: dict = field(default_factory=dict)


def relative_pose_front(match, kp_q, kp_t, view_q, view_t, E_hyp, thr_sq, ratio=0.75, want_points=False,
                        n_samples=410, seed=0, decompose="device", item_key=None, want_counts=False,
                        refine=True, noalloc=False) -> Front:
    """Everything up to the inlier mask, plus (decompose="host") an asynchronous copy of the winning E of every pair.

    match: callable returning (idx, d1, d2, kind) with kind "l2" or "hamming",
           e.g. ``lambda: (*matching.match_l2(...), "l2")`` — or (idx, keep, "l2_keep") when the ratio test was already
           applied at `ratio` by the matcher (``matching.match_l2_ratio``: same idx and keep, fewer exact re-scores).
    kp_q/kp_t: [V, n, 2] normalised keypoints; view_q/view_t: int32 [B] (device) or None.
    E_hyp: [B, H, 9] float32 device tensor of essential-matrix hypotheses, or None: then they are
           generated on the device from `n_samples` Philox-drawn minimal samples of the pair's own
           matches by the 5-point solver (10 hypothesis slots per sample, H = 10*n_samples; a slot holds a model
           only for a real root — ``want_counts`` puts the per-sample numbers of real models, int32 [B, n_samples],
           into ``Front.extra["n_solutions"]``).
    item_key: per-pair Philox keys (int tensor [B], e.g. global pair ids) so that a pair draws the same minimal
           samples in whatever batch slot, chunk or rank it is processed; None = position in the batch.
    noalloc: the 5-point solver keeps its records in the library's static scratch (``ph_solve_essential_5pt_noalloc``:
           at most 8192 samples per batch, one batch at a time per GPU) — for callers under graph capture (plan.py).
    """
    r = match()
    if r[-1] == "l2_keep":
        idx, keep = r[0], r[1]
    else:
        idx, d1, d2, kind = r
        keep = (matching.ratio_filter_l2 if kind == "l2" else matching.ratio_filter_hamming)(idx, d1, d2, ratio)
    m = matching.compact_matches(idx, keep, kp_q, kp_t, view_q, view_t, want_indices=False)
    pts = (m["x1"], m["y1"], m["x2"], m["y2"])
    extra = {}
    front_opts = dict(thr_sq=float(thr_sq), refine=bool(refine))
    generated = E_hyp is None
    if E_hyp is None:
        samples = ransac.sample_minimal(seed, m["n_match"], pts[0].shape[1], n_samples, item_key=item_key)
        E_hyp, n_sol = ransac.solve_essential_5pt(*pts, samples, want_counts=True, noalloc=noalloc)
        if want_counts:
            extra["n_solutions"] = n_sol
    if generated:
        # about half of the solver's slots are empty: score the live models packed together (a wave of eight empty
        # slots leaves at once), then map the winner's compact position back to its slot — same winner, ties included
        Ec, slot_map, extra["n_live"] = ransac.compact_models(E_hyp, counts=n_sol)    # the solver's counts place the pieces: many blocks per pair
        cnt = ransac.score_essential(Ec, *pts, thr_sq, m["n_match"], bailout=True)   # only the winner is read
        bi, bc = ransac.select_best(cnt, slot_map)                        # the winner's slot: arg-max + slot lookup in one launch
    else:
        cnt = ransac.score_essential(E_hyp, *pts, thr_sq, m["n_match"], bailout=True)    # only the winner is read
        bi, bc = ransac.select_best(cnt)
    mask = ransac.inlier_mask_essential(E_hyp, bi, *pts, thr_sq, m["n_match"])
    if decompose == "device":
        return Front(pts, m["n_match"], E_hyp, bi, bc, mask, None, None, want_points, extra, front_opts)
    if decompose != "host":
        raise ValueError("decompose must be 'device' or 'host'")
    B = E_hyp.shape[0]
    E_best = ransac.gather_models(E_hyp, bi)
    host = torch.empty((B, 9), dtype=torch.float32, pin_memory=True)
    host.copy_(E_best, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return Front(pts, m["n_match"], E_hyp, bi, bc, mask, host, ev, want_points, extra, front_opts)


def relative_pose_back(f: Front):
    """Decomposition of the winning E matrices (on the host when the front copied them there, else on the device),
    4-candidate triangulation + cheirality vote, pose table."""
    dev = f.E_hyp.device
    B = f.E_hyp.shape[0]
    if f.E_best_host is None:
        P1d, P2d = ransac.decompose_essential(f.E_hyp, f.best_idx)
    else:
        f.copied.synchronize()
        P1, P2, _, _ = geometry.candidate_projections(f.E_best_host.numpy().reshape(B, 3, 3))
        P1d = torch.from_numpy(P1).to(dev, non_blocking=True)
        P2d = torch.from_numpy(P2).to(dev, non_blocking=True)
    xyz, front = triangulation.triangulate(P1d, P2d, *f.pts, n_valid=f.n_match, mask=f.mask,
                                           want_points=f.want_points, want_front=True)
    table, win = triangulation.pose_table(P2d, front, f.best_count, f.n_match, f.best_idx, min_inliers=5)
    if f.opts.get("refine", False):
        # the cheirality winner, refined on all the pair's correspondences under a loss truncated at 3 x the RANSAC
        # threshold (north_star's "per-hypothesis refinement", here on the device: no host round trip)
        ransac.refine_relative_pose(*f.pts, table, REFINE_TRUNC * REFINE_TRUNC * f.opts["thr_sq"], n_valid=f.n_match,
                                    max_iters=REFINE_ITERS)
    out = dict(table=table, E_hyp=f.E_hyp, n_match=f.n_match, best_idx=f.best_idx, best_count=f.best_count,
               mask=f.mask, front=front, win=win, **f.extra)
    if f.want_points:
        if f.opts.get("refine", False):
            # the points of the REFINED pose (the table's R | t), over the winner's inlier mask
            P2r = torch.cat([table[:, 0:9].reshape(B, 3, 3), table[:, 9:12].reshape(B, 3, 1)], dim=2).reshape(B, 1, 12).contiguous()
            out["xyz"] = triangulation.triangulate(P1d[:, :1].contiguous(), P2r, *f.pts, n_valid=f.n_match, mask=f.mask)[0][:, 0]
        else:
            out["xyz"] = xyz[torch.arange(B, device=dev), win.long()]
        out["pts"] = f.pts
    return out


def relative_pose_batch(match, kp_q, kp_t, view_q, view_t, E_hyp, thr_sq, ratio=0.75, want_points=False,
                        n_samples=410, seed=0, decompose="device", item_key=None, want_counts=False, refine=True,
                        noalloc=False):
    """front + back for one batch of B pairs.  Returns dict(table [B,16] device f32, n_match, best_idx,
    best_count, mask, E_hyp, front, win and, with want_points, xyz / pts; when the hypotheses were generated here also
    n_live int32 [B], the number of real models of every pair, and with want_counts n_solutions [B, n_samples]).

    With refine=True the table's R | t are the LM-refined pose; best_idx, best_count (table column 12), mask, front and
    win still describe the winning minimal-sample hypothesis the refinement started from (its RANSAC support), and xyz
    are the points of the refined pose over that hypothesis' inlier mask.  best_idx is -1 for a pair without a live
    hypothesis (fewer than 5 matches; status 0)."""
    return relative_pose_back(relative_pose_front(match, kp_q, kp_t, view_q, view_t, E_hyp, thr_sq, ratio,
                                                  want_points, n_samples, seed, decompose, item_key, want_counts,
                                                  refine, noalloc))


def absolute_pose_batch(X, Y, Z, u, v, K4, thr_px, n_valid=None, n_samples=4096, seed=0, refine=True, max_iters=20,
                        want_counts=False):
    """RANSAC absolute pose (PnP) of B frames from 2-D/3-D correspondences, all on the device (SURVEY.md §8d cfg 5):

        Philox 3-point samples -> P3P (4 hypothesis slots per sample) -> score H = 4*n_samples hypotheses
          -> best hypothesis -> inlier mask -> Levenberg-Marquardt on the inliers (the reference's solver
          semantics, main.py:101-104) -> final inlier count.

    X, Y, Z, u, v: float32 ``[B, N]`` SoA rows, (u, v) undistorted pixels; K4 = (fx, fy, cx, cy) float64 device
    tensor.  Returns dict(pose [B,6] f64 (rvec, tvec), pose0 (the winning minimal hypothesis), best_count,
    best_idx (slot), count [B,H] (of P_compact's order: exact for the winner, lower bounds for abandoned hypotheses), mask, rms,
    iters, P_hyp (slot layout), P_compact / slot_map (the live models packed, and the slot each came from))."""
    fx, fy, cx, cy = (K4[k].to(torch.float32) for k in range(4))
    xn, yn = (u - cx) / fx, (v - cy) / fy
    B, N = X.shape
    thr_sq = float(thr_px) ** 2
    samples = ransac.sample_minimal(seed, n_valid, N, n_samples, m=3) if n_valid is not None else \
        ransac.sample_minimal(seed, torch.full((B,), N, dtype=torch.int32, device=X.device), N, n_samples, m=3)
    # n_solutions [B, n_samples]: real models per minimal sample (a slot holds a model only for a solution)
    P, n_sol = ransac.solve_p3p(X, Y, Z, xn, yn, samples, K4=K4, want_counts=True)
    Pc, slot_map, _ = ransac.compact_models(P, counts=n_sol)                           # live models packed (see relative_pose_front)
    if not want_counts:
        n_sol = None
    cnt = ransac.score_pnp(Pc, X, Y, Z, u, v, thr_sq, n_valid, bailout=True)          # only the winner is read; compact order
    bi, bc = ransac.select_best(cnt, slot_map)
    mask = ransac.inlier_mask_pnp(P, bi, X, Y, Z, u, v, thr_sq, n_valid)
    pose0 = ransac.pose_from_projection(P, bi, K4)
    out = dict(pose0=pose0, best_count=bc, best_idx=bi, mask=mask, P_hyp=P, count=cnt, P_compact=Pc, slot_map=slot_map)
    if n_sol is not None:
        out["n_solutions"] = n_sol
    if refine:
        pose, rms, iters = ransac.refine_pose_lm_soa(X, Y, Z, u, v, pose0, K4, None, mask, n_valid, max_iters)
        out.update(pose=pose, rms=rms, iters=iters)
    else:
        out.update(pose=pose0)
    return out
