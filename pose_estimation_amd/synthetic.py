"""Seeded synthetic scenes for the configurations of BASELINE.json.

The reference ships no images (its .gitignore excludes every image type) and no data, so every
workload is generated: NumPy on the host, PCG64(seed), seed = 1000 + config number unless a
caller says otherwise.  This module only makes INPUTS (descriptors, keypoints, poses); nothing
here is on the measured path.
"""
from __future__ import annotations

import numpy as np

# the reference's intrinsics (main.py:28-34) — used to turn pixel noise into normalised units
FX, FY, CX, CY = 3108.95, 3101.48, 1129.30, 1940.89
DIST = np.array([0.148, -0.510, 0.003, -0.0004, 0.635])
BOARD_W, BOARD_H = 0.375, 0.235  # main.py:476-477


def rng_for(config: int, seed: int | None = None) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(1000 + config if seed is None else seed))


# ------------------------------------------------------------------ small geometry helpers
def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def look_at(C, target, up=(0.0, 0.0, 1.0)):
    """World->camera rotation of a camera at C looking at target (camera +Z forward, +Y down)."""
    C, target, up = (np.asarray(v, np.float64) for v in (C, target, up))
    z = target - C
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    if np.linalg.norm(x) < 1e-9:
        x = np.cross(z, np.array([0.0, 1.0, 0.0]))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z])  # rows = camera axes in world coordinates


def skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


def essential_from_pose(R, t):
    """E with x2' E x1 = 0 for X2 = R X1 + t."""
    return skew(t / np.linalg.norm(t)) @ R


def relative_pose(R1, t1, R2, t2):
    """(R, t) with X2 = R X1 + t from two world->camera poses (main.py:454-463 geometry)."""
    R = R2 @ R1.T
    return R, t2 - R @ t1


def ring_cameras(n, radius, height, target, phase=0.0):
    """n world->camera poses on a horizontal ring around `target`, all looking at it."""
    Rs, ts = [], []
    for k in range(n):
        a = phase + 2 * np.pi * k / n
        C = np.asarray(target, np.float64) + np.array([radius * np.cos(a), radius * np.sin(a), height])
        R = look_at(C, target)
        Rs.append(R)
        ts.append(-R @ C)
    return np.stack(Rs), np.stack(ts)


# ------------------------------------------------------------------ descriptors
def sift_like(rng, n, dim=128):
    """Unit-norm, non-negative, clipped at 0.2 and renormalised (SIFT's own normalisation)."""
    d = np.abs(rng.standard_normal((n, dim)))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.minimum(d, 0.2)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return d.astype(np.float32)


def perturb_sift(rng, d, sigma):
    p = d.astype(np.float64) + rng.normal(0.0, sigma, d.shape)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    return p.astype(np.float32)


def flip_bits(rng, codes, p):
    """codes: uint8 [n, 32]; flip each of the 256 bits with probability p."""
    flips = rng.random((codes.shape[0], codes.shape[1] * 8)) < p
    return codes ^ np.packbits(flips, axis=1, bitorder="little")


# ------------------------------------------------------------------ config 2: one two-view pair
def two_view_pair(n=8192, dim=128, planted=0.6, sigma_desc=0.05, noise_px=1.0, seed=None, config=2):
    """Two views of a 2 m cube of points 4 m ahead; 0.5 m baseline, 10 degrees of yaw.

    Returns a dict: desc1/desc2 float32 [n, dim]; kp1/kp2 float32 [n, 2] in NORMALISED image
    coordinates (pixel noise / fx); partner [n] (index into view 2 of the planted match of
    query i, -1 when query i is an outlier); R, t, E ground truth (X2 = R X1 + t).
    """
    rng = rng_for(config, seed)
    n_pl = int(round(planted * n))
    R = rot_y(np.deg2rad(-10.0))
    C2 = np.array([0.5, 0.0, 0.0])
    t = -R @ C2
    X = rng.uniform(-1.0, 1.0, (n_pl, 3)) + np.array([0.0, 0.0, 4.0])
    sig = noise_px / FX

    def proj(Xw, Rm, tv):
        Xc = Xw @ Rm.T + tv
        return Xc[:, :2] / Xc[:, 2:3]

    kp1 = np.empty((n, 2)); kp2 = np.empty((n, 2))
    perm = rng.permutation(n)                       # where view-2 copies of the planted points live
    partner = np.full(n, -1, np.int64)
    partner[:n_pl] = perm[:n_pl]
    kp1[:n_pl] = proj(X, np.eye(3), np.zeros(3)) + rng.normal(0, sig, (n_pl, 2))
    kp2[perm[:n_pl]] = proj(X, R, t) + rng.normal(0, sig, (n_pl, 2))
    # unrelated points in each view
    kp1[n_pl:] = proj(rng.uniform(-1, 1, (n - n_pl, 3)) + [0, 0, 4.0], np.eye(3), np.zeros(3))
    kp2[perm[n_pl:]] = proj(rng.uniform(-1, 1, (n - n_pl, 3)) + [0, 0, 4.0], R, t)

    desc2 = sift_like(rng, n, dim)
    desc1 = sift_like(rng, n, dim)
    desc1[:n_pl] = perturb_sift(rng, desc2[perm[:n_pl]], sigma_desc)
    # shuffle the queries so planted ones are not the first rows
    qperm = rng.permutation(n)
    return dict(desc1=desc1[qperm], desc2=desc2, kp1=kp1[qperm].astype(np.float32), kp2=kp2.astype(np.float32),
                partner=partner[qperm], R=R, t=t, E=essential_from_pose(R, t))


# ------------------------------------------------------------------ config 3/4: binary multi-view
def binary_views(n_views, n_feat, n_landmarks=None, p_flip=0.05, seed=None, config=3, radius=3.0):
    """ORB-like views of one landmark set.

    Returns desc uint8 [V, n_feat, 32], kp float32 [V, n_feat, 2] (normalised), landmark ids
    int32 [V, n_feat], poses (R [V,3,3], t [V,3]) world->camera on a ring of `radius` looking at
    the origin (for > 100 views: rings of 100 stacked in height).
    """
    rng = rng_for(config, seed)
    L = n_landmarks or 2 * n_feat
    codes = rng.integers(0, 256, (L, 32), dtype=np.uint8)
    Xw = rng.uniform(-1.0, 1.0, (L, 3))
    Rs, ts = [], []
    per_ring = 100
    for v0 in range(0, n_views, per_ring):
        nv = min(per_ring, n_views - v0)
        Rr, tr = ring_cameras(nv, radius, 0.5 + 0.1 * (v0 // per_ring), (0.0, 0.0, 0.0), phase=0.013 * v0)
        Rs.append(Rr); ts.append(tr)
    Rs, ts = np.concatenate(Rs), np.concatenate(ts)
    desc = np.empty((n_views, n_feat, 32), np.uint8)
    kp = np.empty((n_views, n_feat, 2), np.float32)
    ids = np.empty((n_views, n_feat), np.int32)
    for v in range(n_views):
        sel = rng.permutation(L)[:n_feat]
        ids[v] = sel
        desc[v] = flip_bits(rng, codes[sel], p_flip)
        Xc = Xw[sel] @ Rs[v].T + ts[v]
        kp[v] = (Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, 1.0 / FX, (n_feat, 2))).astype(np.float32)
    return dict(desc=desc, kp=kp, ids=ids, R=Rs, t=ts)


def float_views(n_views, n_feat, n_landmarks=None, sigma_desc=0.05, seed=None, config=4, radius=3.0, dim=128):
    """SIFT-like float views of one landmark set: the f32 variant of `binary_views` (same cameras, same
    observation model; a view sees a random subset of the landmarks, each descriptor perturbed by
    N(0, sigma_desc) and renormalised).  Returns desc float32 [V, n_feat, dim], kp, ids, R, t."""
    rng = rng_for(config, seed)
    L = n_landmarks or 2 * n_feat
    codes = sift_like(rng, L, dim)
    Xw = rng.uniform(-1.0, 1.0, (L, 3))
    Rs, ts = [], []
    per_ring = 100
    for v0 in range(0, n_views, per_ring):
        nv = min(per_ring, n_views - v0)
        Rr, tr = ring_cameras(nv, radius, 0.5 + 0.1 * (v0 // per_ring), (0.0, 0.0, 0.0), phase=0.013 * v0)
        Rs.append(Rr); ts.append(tr)
    Rs, ts = np.concatenate(Rs), np.concatenate(ts)
    desc = np.empty((n_views, n_feat, dim), np.float32)
    kp = np.empty((n_views, n_feat, 2), np.float32)
    ids = np.empty((n_views, n_feat), np.int32)
    for v in range(n_views):
        sel = rng.permutation(L)[:n_feat]
        ids[v] = sel
        d = codes[sel] + rng.standard_normal((n_feat, dim), dtype=np.float32) * np.float32(sigma_desc)
        np.maximum(d, 0.0, out=d)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        desc[v] = d
        Xc = Xw[sel] @ Rs[v].T + ts[v]
        kp[v] = (Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, 1.0 / FX, (n_feat, 2))).astype(np.float32)
    return dict(desc=desc, kp=kp, ids=ids, R=Rs, t=ts)


def all_pairs(n_views):
    i, j = np.triu_indices(n_views, k=1)
    return i.astype(np.int32), j.astype(np.int32)


# ------------------------------------------------------------------ config 5: PnP stress frames
def pnp_frames(B=32, n=65536, outliers=0.3, noise_px=1.0, seed=None, config=5):
    """B frames of n 2D-3D correspondences; `outliers` of the 2-D points replaced by uniform pixels.

    Returns X,Y,Z,u,v float32 [B, n] (u,v in pixels, undistorted), R [B,3,3], t [B,3] ground
    truth, inlier_truth bool [B, n].  Image size follows the reference's principal point.
    """
    rng = rng_for(config, seed)
    W, Himg = 2.0 * CX, 2.0 * CY
    out = {k: np.empty((B, n), np.float32) for k in "XYZuv"}
    Rs, ts = np.empty((B, 3, 3)), np.empty((B, 3))
    truth = np.ones((B, n), bool)
    for b in range(B):
        Xw = rng.uniform(-1.0, 1.0, (n, 3))
        a = rng.uniform(0, 2 * np.pi)
        C = np.array([4.0 * np.cos(a), 4.0 * np.sin(a), rng.uniform(-0.5, 0.5)])
        R = look_at(C, (0.0, 0.0, 0.0))
        t = -R @ C
        Xc = Xw @ R.T + t
        u = FX * Xc[:, 0] / Xc[:, 2] + CX + rng.normal(0, noise_px, n)
        v = FY * Xc[:, 1] / Xc[:, 2] + CY + rng.normal(0, noise_px, n)
        bad = rng.random(n) < outliers
        u[bad] = rng.uniform(0, W, bad.sum()); v[bad] = rng.uniform(0, Himg, bad.sum())
        truth[b] = ~bad
        out["X"][b], out["Y"][b], out["Z"][b] = Xw[:, 0], Xw[:, 1], Xw[:, 2]
        out["u"][b], out["v"][b] = u, v
        Rs[b], ts[b] = R, t
    out.update(R=Rs, t=ts, inlier_truth=truth)
    return out


def projection_matrix(R, t, pixels=True):
    """Row-major 12-vector of K[R|t] (pixels) or [R|t] (normalised)."""
    Rt = np.concatenate([np.asarray(R, np.float64), np.asarray(t, np.float64).reshape(3, 1)], axis=1)
    if pixels:
        K = np.array([[FX, 0, CX], [0, FY, CY], [0, 0, 1.0]])
        Rt = K @ Rt
    return Rt.reshape(12)


def perturbed_poses(rng, R, t, H, rot_sigma=0.02, t_sigma=0.05):
    """H pose hypotheses around (R, t): hypothesis 0 is the truth, the rest are perturbed."""
    Ps = np.empty((H, 3, 4))
    for h in range(H):
        if h == 0:
            Rh, th = R, t
        else:
            w = rng.normal(0, rot_sigma, 3)
            ang = np.linalg.norm(w)
            Kx = skew(w / ang) if ang > 0 else np.zeros((3, 3))
            dR = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
            Rh, th = dR @ R, t + rng.normal(0, t_sigma, 3)
        Ps[h, :, :3], Ps[h, :, 3] = Rh, th
    return Ps


# ------------------------------------------------------------------ config 1 substitute: board views
def board_object_points():
    """GridBoard((11,7), 0.025, 0.010) corners [77, 4, 3] float32, id = j*11+i (validate_pose.py:162-176)."""
    m, s = 0.025, 0.010
    pts = np.zeros((77, 4, 3), np.float32)
    for j in range(7):
        for i in range(11):
            x0, y0 = i * (m + s), j * (m + s)
            pts[j * 11 + i] = [[x0, y0, 0], [x0 + m, y0, 0], [x0 + m, y0 + m, 0], [x0, y0 + m, 0]]
    return pts


def distort_project(Xc):
    """Pinhole + 5-coefficient distortion of camera-frame points with the reference's K, dist."""
    x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
    k1, k2, p1, p2, k3 = DIST
    r2 = x * x + y * y
    g = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * g + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * g + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([FX * xd + CX, FY * yd + CY], axis=1)


def board_views(n_views=8, radius=0.6, height=0.4, noise_px=0.3, seed=None, config=1, drop=0.0):
    """Synthetic detections of the reference's board from a ring of cameras (SURVEY.md §8d cfg 1).

    The camera 'up' is chosen so the board's +Z faces the cameras.  Returns a list of dicts with
    the shapes cv2.aruco.detectMarkers hands back: corners = tuple of (1,4,2) float32,
    ids = (n,1) int32, plus the ground-truth R, t (board -> camera).
    """
    rng = rng_for(config, seed)
    obj = board_object_points()
    centre = np.array([BOARD_W / 2, BOARD_H / 2, 0.0])
    views = []
    for k in range(n_views):
        a = 2 * np.pi * k / n_views + 0.1
        C = centre + np.array([radius * np.cos(a), radius * np.sin(a), height])
        R = look_at(C, centre)
        t = -R @ C
        ids = np.arange(77)
        if drop > 0:
            ids = ids[rng.random(77) >= drop]
        ids = rng.permutation(ids)              # detectors return markers in no particular order
        corners = []
        for m in ids:
            uv = distort_project(obj[m].astype(np.float64) @ R.T + t) + rng.normal(0, noise_px, (4, 2))
            corners.append(uv.astype(np.float32).reshape(1, 4, 2))
        views.append(dict(filename=f"synthetic_{k:02d}.png", corners=tuple(corners),
                          ids=ids.astype(np.int32).reshape(-1, 1), R=R, t=t))
    return views
