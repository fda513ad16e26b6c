"""CPU restatement (NumPy, float64) of the relative-pose refinement `ph_refine_relative_pose`.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): the product path never
imports this module.

Parity status: **parity unpinned** — the reference has no two-view code at all; the nearest thing it does is
"refine after the initial solve" inside cv2.solvePnP(..., SOLVEPNP_ITERATIVE) (/root/reference/main.py:101-104),
which is Levenberg-Marquardt on the reprojection error of ONE view.  north_star leaves "the per-hypothesis
refinement" to the implementation; this file and the device kernel state the same algorithm twice:

    minimise  sum_i  min(r_i^2, c^2),   r_i = (x2_i' E x1_i) / sqrt((E x1)_0^2 + (E x1)_1^2 + (E' x2)_0^2 + (E' x2)_1^2)
    over the 5 degrees of freedom of E = [t]x R, |t| = 1, on the correspondences whose `valid` bit is set (all when
    none is given): the Sampson residual of DESIGN.md §4, here as a signed distance, in float64, under a truncated
    quadratic (MSAC) loss — a correspondence takes part in the normal equations of a pass iff r_i^2 < c^2 at the
    pose of that pass, so the inlier set follows the pose instead of staying the minimal-sample winner's (whose
    tight, biased set holds the refinement back: 0.53 deg after refining on it vs 0.005 deg, tests/test_refine_e.py);
    update: R <- exp([dw]x) R (3 dof), t <- normalise(t + d3 b1 + d4 b2) with (b1, b2) a basis of t's tangent
    plane (2 dof); damping: Marquardt (lambda * diag(J'J)), lambda0 = 1e-3, /10 on success, x10 on failure.
"""
from __future__ import annotations

import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def exp_so3(w):
    th = float(np.linalg.norm(w))
    if th < 1e-12:
        return np.eye(3) + skew(w)
    k = np.asarray(w, np.float64) / th
    K = skew(k)
    return np.eye(3) + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K)


def tangent_basis(t):
    """(b1, b2): b1 = normalise(t x e_a) with a the axis of the smallest |t_a| (lowest index on ties), b2 = t x b1."""
    a = int(np.argmin(np.abs(t)))
    e = np.zeros(3)
    e[a] = 1.0
    b1 = np.cross(t, e)
    b1 /= np.linalg.norm(b1)
    return b1, np.cross(t, b1)


def _sums(R, t, x1, y1, x2, y2, c_sq):
    """J'J [5,5], J'r [5], truncated cost, sum of r^2 and number of the correspondences with r^2 < c_sq at (R, t)."""
    E = skew(t) @ R
    b1, b2 = tangent_basis(t)
    G = [skew(t) @ skew(np.eye(3)[k]) @ R for k in range(3)] + [skew(b1) @ R, skew(b2) @ R]
    X1 = np.stack([x1, y1, np.ones_like(x1)])          # [3, n]
    X2 = np.stack([x2, y2, np.ones_like(x2)])
    Ex1 = E @ X1
    Etx2 = E.T @ X2
    num = np.sum(X2 * Ex1, axis=0)
    den = Ex1[0] ** 2 + Ex1[1] ** 2 + Etx2[0] ** 2 + Etx2[1] ** 2
    den = np.maximum(den, 1e-300)
    isq = 1.0 / np.sqrt(den)
    r = num * isq
    w = (r * r < c_sq)
    J = np.empty((5, len(x1)))
    for k in range(5):
        Gx1 = G[k] @ X1
        Gtx2 = G[k].T @ X2
        dnum = np.sum(X2 * Gx1, axis=0)
        dden = 2.0 * (Ex1[0] * Gx1[0] + Ex1[1] * Gx1[1] + Etx2[0] * Gtx2[0] + Etx2[1] * Gtx2[1])
        J[k] = dnum * isq - 0.5 * num * dden * isq / den
    J = J * w
    rw = r * w
    ssq, nin = float(rw @ rw), int(w.sum())
    return J @ J.T, J @ rw, ssq + (len(x1) - nin) * float(c_sq), ssq, nin


def refine_relative_pose(R0, t0, x1, y1, x2, y2, c_sq, valid=None, max_iters=10):
    """LM refinement of (R, t), |t| = 1, under the truncated loss min(r^2, c_sq) over the `valid` correspondences.

    Returns (R, t, rms over the final inliers, number of final inliers, iters); with fewer than 6 correspondences
    the input is returned unchanged (iters = 0)."""
    sel = np.ones(len(x1), bool) if valid is None else np.asarray(valid, bool)
    x1, y1, x2, y2 = (np.asarray(a, np.float64)[sel] for a in (x1, y1, x2, y2))
    R = np.asarray(R0, np.float64).copy()
    t = np.asarray(t0, np.float64).copy()
    n = len(x1)
    nt = np.linalg.norm(t)
    if n < 6 or not (nt > 0):
        return R, t, 0.0, 0, 0
    t /= nt
    A, g, cost, ssq, nin = _sums(R, t, x1, y1, x2, y2, c_sq)
    lam, it = 1e-3, 0
    while it < max_iters:
        it += 1
        M = A + lam * np.diag(np.diag(A)) + 1e-300 * np.eye(5)
        try:
            L = np.linalg.cholesky(M)
        except np.linalg.LinAlgError:
            lam *= 10.0
            if lam > 1e12:
                break
            continue
        d = -np.linalg.solve(L.T, np.linalg.solve(L, g))
        b1, b2 = tangent_basis(t)
        Rn = exp_so3(d[:3]) @ R
        tn = t + d[3] * b1 + d[4] * b2
        tn /= np.linalg.norm(tn)
        An, gn, cn, ssqn, ninn = _sums(Rn, tn, x1, y1, x2, y2, c_sq)
        if cn < cost:
            done = (cost - cn) <= 1e-9 * max(cost, 1e-300) or float(d @ d) < 1e-20
            R, t, A, g, cost, ssq, nin = Rn, tn, An, gn, cn, ssqn, ninn
            lam = max(lam * 0.1, 1e-12)
            if done:
                break
        else:
            lam *= 10.0
            if lam > 1e12:
                break
    return R, t, float(np.sqrt(ssq / nin)) if nin else 0.0, nin, it
