"""NumPy P3P checker.  TEST INFRASTRUCTURE ONLY (see oracle/oracle.py header).

No reference counterpart (the reference calls cv2.solvePnP on all points, main.py:101-104; north_star adds
RANSAC with P3P hypotheses).  Independent of the library's root finder and polishing: the quartic in v = s3/s1 is
built from the three law-of-cosines equations with numpy.polynomial arithmetic, solved by np.roots (companion
matrix eigenvalues), and each real positive root is turned into a pose by a Kabsch alignment (SVD) instead of the
library's triangle frames.  A pose is a P3P solution iff R is a rotation, depths are positive and the three points
reproject onto their bearings — `check_solution` states exactly that and the tests apply it to every library output.
"""
from __future__ import annotations

import numpy as np
from numpy.polynomial import polynomial as Pl


def bearings(xy):
    xy = np.asarray(xy, np.float64).reshape(-1, 2)
    f = np.concatenate([xy, np.ones((len(xy), 1))], axis=1)
    return f / np.linalg.norm(f, axis=1, keepdims=True)


def solve_p3p(X, f):
    """X [3,3] world points, f [3,3] unit bearings -> list of (R, t) with x_cam = R X + t."""
    X = np.asarray(X, np.float64); f = np.asarray(f, np.float64)
    a2 = np.sum((X[1] - X[2]) ** 2); b2 = np.sum((X[0] - X[2]) ** 2); c2 = np.sum((X[0] - X[1]) ** 2)
    ca, cb, cg = f[1] @ f[2], f[0] @ f[2], f[0] @ f[1]
    Q = np.array([1.0, -2 * cb, 1.0])
    N = Pl.polysub((a2 - c2) * Q, b2 * np.array([-1.0, 0.0, 1.0]))
    D = 2 * b2 * np.array([cg, -ca])
    D2 = Pl.polymul(D, D)
    quartic = Pl.polysub(b2 * Pl.polyadd(Pl.polyadd(D2, Pl.polymul(N, N)), -2 * cg * Pl.polymul(N, D)),
                         c2 * Pl.polymul(Q, D2))
    out = []
    for v in np.roots(quartic[::-1]):
        if abs(v.imag) > 1e-6 * max(1.0, abs(v.real)) or v.real <= 0:
            continue
        v = v.real
        Dv = Pl.polyval(v, D)
        if abs(Dv) < 1e-12 * b2:
            continue
        u = Pl.polyval(v, N) / Dv
        Qv = Pl.polyval(v, Q)
        if u <= 0 or Qv <= 0:
            continue
        s1 = np.sqrt(b2 / Qv)
        Y = np.array([s1, u * s1, v * s1])[:, None] * f
        # Kabsch: rotation aligning the centred triangles
        Xc, Yc = X - X.mean(0), Y - Y.mean(0)
        U, _, Vt = np.linalg.svd(Yc.T @ Xc)
        S = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
        R = U @ S @ Vt
        t = Y.mean(0) - R @ X.mean(0)
        if np.abs(R @ X.T + t[:, None] - Y.T).max() > 1e-7 * np.sqrt(a2 + b2 + c2):
            continue   # inconsistent triangle (spurious root)
        out.append((R, t))
    return out


def check_solution(R, t, X, f):
    """(orthonormality error, det, min depth, max bearing mismatch) of a claimed solution."""
    Y = (R @ np.asarray(X, np.float64).T).T + t
    d = np.linalg.norm(Y, axis=1)
    return (np.abs(R @ R.T - np.eye(3)).max(), np.linalg.det(R), float((Y * f).sum(1).min()),
            float(np.abs(Y / d[:, None] - f).max()))
