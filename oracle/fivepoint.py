"""NumPy float64 restatement of the 5-point essential-matrix minimal solver.  TEST INFRASTRUCTURE ONLY.

There is no reference code for this (the reference only calls cv2.solvePnP); the algorithm is
Nistér's ("An efficient solution to the five-point relative pose problem", PAMI 2004):
null space of the 5x9 epipolar system -> 10 cubic constraints in (x, y, z) -> Gauss-Jordan ->
3x3 polynomial matrix in z -> degree-10 determinant -> real roots -> E.  Batched over samples.
Root finding here is numpy.roots (companion-matrix eigenvalues) — deliberately NOT the Aberth
iteration the device kernel uses, so the two only share the published algebra.
"""
from __future__ import annotations

import numpy as np

MUL11 = np.array([[0, 1, 2, 6], [1, 3, 4, 7], [2, 4, 5, 8], [6, 7, 8, 9]])
MUL21 = np.array([[0, 1, 2, 10], [1, 3, 4, 11], [2, 4, 5, 12], [3, 6, 7, 13], [4, 7, 8, 14], [5, 8, 9, 15],
                  [10, 11, 12, 16], [11, 13, 14, 17], [12, 14, 15, 18], [16, 17, 18, 19]])
NISTER = np.array([0, 6, 1, 3, 2, 10, 7, 13, 4, 11, 5, 12, 16, 8, 14, 17, 9, 15, 18, 19])


def _mul11(a, b):
    out = np.zeros(a.shape[:-1] + (10,))
    for i in range(4):
        for j in range(4):
            out[..., MUL11[i, j]] += a[..., i] * b[..., j]
    return out


def _mul21(a, b):
    out = np.zeros(a.shape[:-1] + (20,))
    for i in range(10):
        for j in range(4):
            out[..., MUL21[i, j]] += a[..., i] * b[..., j]
    return out


def epipolar_rows(x1, y1, x2, y2):
    """Rows q with q . vec(E) = x2' E x1 (E row-major)."""
    o = np.ones_like(x1)
    return np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, o], axis=-1)


def solve(x1, y1, x2, y2):
    """x1.. : [S, 5] float64.  Returns list (len S) of arrays [k, 3, 3] of unit-Frobenius E solutions."""
    Q = epipolar_rows(*(np.asarray(a, np.float64) for a in (x1, y1, x2, y2)))       # [S,5,9]
    S = Q.shape[0]
    _, _, Vt = np.linalg.svd(Q)
    basis = Vt[:, 5:, :]                                                            # [S,4,9]: X, Y, Z, W
    # E entries as linear polynomials in (x,y,z,1): coefficient j of entry (r,c) is basis[j][3r+c]
    Ep = np.transpose(basis, (0, 2, 1)).reshape(S, 3, 3, 4)
    EEt = np.zeros((S, 3, 3, 10))
    for r in range(3):
        for c in range(3):
            for k in range(3):
                EEt[:, r, c] += _mul11(Ep[:, r, k], Ep[:, c, k])
    tr = EEt[:, 0, 0] + EEt[:, 1, 1] + EEt[:, 2, 2]
    eqs = []
    for r in range(3):
        for c in range(3):
            acc = np.zeros((S, 20))
            for k in range(3):
                acc += 2.0 * _mul21(EEt[:, r, k], Ep[:, k, c])
            acc -= _mul21(tr, Ep[:, r, c])
            eqs.append(acc)
    det = (_mul21(_mul11(Ep[:, 0, 1], Ep[:, 1, 2]) - _mul11(Ep[:, 0, 2], Ep[:, 1, 1]), Ep[:, 2, 0]) +
           _mul21(_mul11(Ep[:, 0, 2], Ep[:, 1, 0]) - _mul11(Ep[:, 0, 0], Ep[:, 1, 2]), Ep[:, 2, 1]) +
           _mul21(_mul11(Ep[:, 0, 0], Ep[:, 1, 1]) - _mul11(Ep[:, 0, 1], Ep[:, 1, 0]), Ep[:, 2, 2]))
    A = np.stack([det] + eqs, axis=1)[:, :, NISTER]                                 # [S,10,20]
    out = []
    for s in range(S):
        try:
            Bm = np.linalg.solve(A[s, :, :10], A[s, :, 10:])                        # rows x3..xy | right block
        except np.linalg.LinAlgError:
            out.append(np.zeros((0, 3, 3))); continue
        e, f, g, h, i, j = Bm[4], Bm[5], Bm[6], Bm[7], Bm[8], Bm[9]

        def row(p, q):   # p - z*q as polynomials in z (descending), for the x, y, 1 parts
            px = np.array([-q[0], p[0] - q[1], p[1] - q[2], p[2]])
            py = np.array([-q[3], p[3] - q[4], p[4] - q[5], p[5]])
            p1 = np.array([-q[6], p[6] - q[7], p[7] - q[8], p[8] - q[9], p[9]])
            return px, py, p1
        (kx, ky, k1), (lx, ly, l1), (mx, my, m1) = row(e, f), row(g, h), row(i, j)
        P = np.polynomial.polynomial
        c = lambda a: a[::-1]                                                       # ascending order for np.polynomial
        detz = (P.polymul(c(kx), P.polysub(P.polymul(c(ly), c(m1)), P.polymul(c(l1), c(my)))) -
                P.polymul(c(ky), P.polysub(P.polymul(c(lx), c(m1)), P.polymul(c(l1), c(mx)))) +
                P.polymul(c(k1), P.polysub(P.polymul(c(lx), c(my)), P.polymul(c(ly), c(mx)))))
        detz = np.concatenate([detz, np.zeros(11 - len(detz))])
        roots = np.roots(detz[::-1])
        sols = []
        for z in roots[np.abs(roots.imag) < 1e-9 * np.maximum(1.0, np.abs(roots.real))].real:
            Bz = np.array([[np.polyval(kx, z), np.polyval(ky, z), np.polyval(k1, z)],
                           [np.polyval(lx, z), np.polyval(ly, z), np.polyval(l1, z)],
                           [np.polyval(mx, z), np.polyval(my, z), np.polyval(m1, z)]])
            _, _, vt = np.linalg.svd(Bz)
            v = vt[-1]
            if abs(v[2]) < 1e-14:
                continue
            x, y = v[0] / v[2], v[1] / v[2]
            E = (x * basis[s, 0] + y * basis[s, 1] + z * basis[s, 2] + basis[s, 3]).reshape(3, 3)
            sols.append(E / np.linalg.norm(E))
        out.append(np.array(sols).reshape(-1, 3, 3))
    return out


def essential_residuals(E, x1, y1, x2, y2):
    """(max |x2' E x1| over the 5 points, ||2EE'E - tr(EE')E||, |det E|) for a unit-norm E."""
    h1 = np.stack([x1, y1, np.ones_like(x1)], -1); h2 = np.stack([x2, y2, np.ones_like(x2)], -1)
    epi = np.abs(np.einsum("ni,ij,nj->n", h2, E, h1)).max()
    EEt = E @ E.T
    return epi, np.linalg.norm(2 * EEt @ E - np.trace(EEt) * E), abs(np.linalg.det(E))


def philox4x32_10(counter, key):
    """Philox-4x32-10 (Salmon et al. 2011): counter [..., 4] uint32, key [..., 2] uint32 -> [..., 4] uint32."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
    c = [np.asarray(counter[..., i], np.uint32) for i in range(4)]
    k0, k1 = np.asarray(key[..., 0], np.uint32), np.asarray(key[..., 1], np.uint32)
    for _ in range(10):
        p0 = M0 * c[0].astype(np.uint64); p1 = M1 * c[2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        with np.errstate(over="ignore"):
            k0 = (k0 + W0).astype(np.uint32); k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def sample_minimal(seed, item, n_valid, n_samples, m=5):
    """The library's minimal-sample rule: sample s of item b draws m distinct indices in
    [0, n_valid).  Draw t (t = 0, 1, ...) is Philox4x32-10(counter = (s, t, 0, 0), key = (seed, item)) word 0
    mapped by the multiply-shift  idx = (word * n_valid) >> 32 ; a draw equal to an earlier index of
    the same sample is skipped.  n_valid < m -> all -1."""
    out = np.full((n_samples, m), -1, np.int32)
    if n_valid < m:
        return out
    for s in range(n_samples):
        got, t = [], 0
        while len(got) < m:
            w = philox4x32_10(np.array([s, t, 0, 0], np.uint32), np.array([seed, item], np.uint32))[0]
            idx = int((int(w) * int(n_valid)) >> 32)
            if idx not in got:
                got.append(idx)
            t += 1
        out[s] = got
    return out
