"""Batched two-view relative-pose pipeline: the hot path north_star describes, end to end.

    match (L2 or Hamming, top-2) -> Lowe ratio test -> compact correspondences
      -> score H essential-matrix hypotheses -> best hypothesis -> inlier mask
      -> [host] decompose E into 4 (R, t) candidates
      -> triangulate with all 4 + cheirality vote -> pose table row

The reference's per-image loop (main.py:163-214) is what gets batched: one launch per stage
handles B pairs; the only host round trip is the 3x3 SVD per pair that north_star leaves on
the host (recoverPose-style decomposition).  Output: a [B, 16] float32 pose table
(R row-major 9, t 3, inliers, matches, winning hypothesis, status) — the unit that is gathered
across ranks.
"""
from __future__ import annotations

import numpy as np
import torch

from . import geometry, matching, ransac, triangulation

STATUS_OK = 1.0
STATUS_DEGENERATE = 0.0


def relative_pose_batch(match, kp_q, kp_t, view_q, view_t, E_hyp, thr_sq, ratio=0.75, want_points=False,
                        n_samples=410, seed=0):
    """Run the pipeline for B pairs.

    match: callable returning (idx, d1, d2) and a flag telling which ratio filter applies,
           e.g. ``lambda: (*matching.match_l2(...), "l2")``.
    kp_q/kp_t: [V, n, 2] normalised keypoints; view_q/view_t: int32 [B] (device) or None.
    E_hyp: [B, H, 9] float32 device tensor of essential-matrix hypotheses, or None: then they are
           generated on the device from `n_samples` Philox-drawn minimal samples of the pair's own
           matches by the 5-point solver (10 hypothesis slots per sample, H = 10*n_samples).
    Returns dict(table [B,16] device f32, n_match, best_idx, best_count, mask, xyz (optional)).
    """
    idx, d1, d2, kind = match()
    keep = (matching.ratio_filter_l2 if kind == "l2" else matching.ratio_filter_hamming)(idx, d1, d2, ratio)
    m = matching.compact_matches(idx, keep, kp_q, kp_t, view_q, view_t, want_indices=False)
    pts = (m["x1"], m["y1"], m["x2"], m["y2"])
    if E_hyp is None:
        samples = ransac.sample_minimal(seed, m["n_match"], pts[0].shape[1], n_samples)
        E_hyp = ransac.solve_essential_5pt(*pts, samples)
    cnt = ransac.score_essential(E_hyp, *pts, thr_sq, m["n_match"])
    bi, bc = ransac.select_best(cnt)
    mask = ransac.inlier_mask_essential(E_hyp, bi, *pts, thr_sq, m["n_match"])

    # host: decomposition of the winning E of every pair (3x3 SVDs)
    B = E_hyp.shape[0]
    E_best = E_hyp[torch.arange(B, device=E_hyp.device), bi.long()].reshape(B, 3, 3)
    P1, P2, Rc, tc = geometry.candidate_projections(E_best.cpu().numpy())
    dev = E_hyp.device
    xyz, front = triangulation.triangulate(torch.from_numpy(P1).to(dev), torch.from_numpy(P2).to(dev), *pts,
                                           n_valid=m["n_match"], mask=mask, want_points=want_points, want_front=True)
    win = torch.argmax(front, dim=1)
    win_h = win.cpu().numpy()
    table = np.zeros((B, 16), np.float32)
    table[:, :9] = Rc[np.arange(B), win_h].reshape(B, 9)
    table[:, 9:12] = tc[np.arange(B), win_h]
    table_d = torch.from_numpy(table).to(dev)
    table_d[:, 12] = bc.float()
    table_d[:, 13] = m["n_match"].float()
    table_d[:, 14] = bi.float()
    table_d[:, 15] = torch.where(bc >= 5, STATUS_OK, STATUS_DEGENERATE)
    out = dict(table=table_d, E_hyp=E_hyp, n_match=m["n_match"], best_idx=bi, best_count=bc, mask=mask, front=front, win=win)
    if want_points:
        out["xyz"] = xyz[torch.arange(B, device=dev), win]
        out["pts"] = pts
    return out
