"""CPU oracle of the camera-pose hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (pose_estimation_amd/) never does.

Two parts:
  * ctypes access to oracle/libposehot_oracle.so (posehot_oracle.c): matching, scoring,
    triangulation, projection — plain C loops in the library's documented op order;
  * NumPy float64 restatements of the reference's own Python-level arithmetic, each citing the
    reference lines it follows (board geometry, camera inversion, quaternion order, statistics).

PARITY STATUS: cv2 is not installed and the reference ships no tests, fixtures or data
(SURVEY.md §0, §8c), so everything that happens inside OpenCV (matchImagePoints, solvePnP,
projectPoints, Rodrigues) is restated from OpenCV's documented models: "parity unpinned".
The one pinned convention is the quaternion order/sign of main.py:138-141, checked against
SciPy-generated vectors in tests/golden/ (SciPy is importable; tests/golden/make_golden.py).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libposehot_oracle.so")
_lib = None

NO_INDEX = -1
HAMMING_NONE = 0x7FFFFFFF


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle`")
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_triangulate.restype = C.c_int
        _lib.orc_compact_matches.restype = C.c_int
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(C.c_int(n))


# ---------------------------------------------------------------- (1) matching
def match_hamming_top2(q, t):
    q, t = _c(q, np.uint8), _c(t, np.uint8)
    nq, nt = q.shape[0], t.shape[0]
    idx = np.empty(nq, np.int32); d1 = np.empty(nq, np.int32); d2 = np.empty(nq, np.int32)
    lib().orc_match_hamming_top2(_p(q), _p(t), nq, nt, q.shape[1] if nq else 32, _p(idx), _p(d1), _p(d2))
    return idx, d1, d2


def row_sqnorm(a):
    a = _c(a, np.float32)
    out = np.empty(a.shape[0], np.float32)
    lib().orc_row_sqnorm(_p(a), a.shape[0], a.shape[1], _p(out))
    return out


def match_l2_top2(q, t):
    """The library's f32 contract (fma chain in MFMA operand order)."""
    q, t = _c(q, np.float32), _c(t, np.float32)
    nq, nt = q.shape[0], t.shape[0]
    idx = np.empty(nq, np.int32); d1 = np.empty(nq, np.float32); d2 = np.empty(nq, np.float32)
    lib().orc_match_l2_top2(_p(q), _p(t), nq, nt, q.shape[1], _p(idx), _p(d1), _p(d2))
    return idx, d1, d2


def match_l2_top2_f64(q, t):
    """Independent float64 sum of squared differences."""
    q, t = _c(q, np.float32), _c(t, np.float32)
    nq, nt = q.shape[0], t.shape[0]
    idx = np.empty(nq, np.int32); d1 = np.empty(nq, np.float64); d2 = np.empty(nq, np.float64)
    lib().orc_match_l2_top2_f64(_p(q), _p(t), nq, nt, q.shape[1], _p(idx), _p(d1), _p(d2))
    return idx, d1, d2


def ratio_filter_hamming(idx, d1, d2, ratio):
    idx, d1, d2 = _c(idx, np.int32), _c(d1, np.int32), _c(d2, np.int32)
    keep = np.empty(idx.shape, np.int32)
    lib().orc_ratio_filter_hamming(_p(idx), _p(d1), _p(d2), C.c_int64(idx.size), C.c_float(ratio), _p(keep))
    return keep


def ratio_filter_l2(idx, d1sq, d2sq, ratio):
    idx, d1sq, d2sq = _c(idx, np.int32), _c(d1sq, np.float32), _c(d2sq, np.float32)
    keep = np.empty(idx.shape, np.int32)
    lib().orc_ratio_filter_l2(_p(idx), _p(d1sq), _p(d2sq), C.c_int64(idx.size), C.c_float(ratio), _p(keep))
    return keep


def compact_matches(idx, keep, kp_q, kp_t):
    idx, keep = _c(idx, np.int32), _c(keep, np.int32)
    kp_q, kp_t = _c(kp_q, np.float32), _c(kp_t, np.float32)
    nq = idx.shape[0]
    out = {k: np.zeros(nq, np.float32) for k in ("x1", "y1", "x2", "y2")}
    mq = np.zeros(nq, np.int32); mt = np.zeros(nq, np.int32)
    n = lib().orc_compact_matches(_p(idx), _p(keep), _p(kp_q), _p(kp_t), nq, _p(out["x1"]), _p(out["y1"]),
                                  _p(out["x2"]), _p(out["y2"]), _p(mq), _p(mt))
    out.update(match_q=mq, match_t=mt, n_match=n)
    return out


# ---------------------------------------------------------------- (2) scoring
def score_essential(E, x1, y1, x2, y2, thr_sq):
    E = _c(E, np.float32).reshape(-1, 9)
    a = [_c(v, np.float32) for v in (x1, y1, x2, y2)]
    count = np.empty(E.shape[0], np.int32)
    lib().orc_score_essential(_p(E), *map(_p, a), E.shape[0], a[0].size, C.c_float(thr_sq), _p(count))
    return count


def score_pnp(P, X, Y, Z, u, v, thr_sq):
    P = _c(P, np.float32).reshape(-1, 12)
    a = [_c(w, np.float32) for w in (X, Y, Z, u, v)]
    count = np.empty(P.shape[0], np.int32)
    lib().orc_score_pnp(_p(P), *map(_p, a), P.shape[0], a[0].size, C.c_float(thr_sq), _p(count))
    return count


def select_best(count):
    count = _c(count, np.int32)
    bi = C.c_int32(); bc = C.c_int32()
    lib().orc_select_best(_p(count), count.size, C.byref(bi), C.byref(bc))
    return bi.value, bc.value


def inlier_mask_essential(E, x1, y1, x2, y2, thr_sq, N=None):
    E = _c(E, np.float32).reshape(9)
    a = [_c(v, np.float32) for v in (x1, y1, x2, y2)]
    n = a[0].size
    N = n if N is None else N
    mask = np.zeros((N + 63) // 64, np.uint64)
    lib().orc_inlier_mask_essential(_p(E), *map(_p, a), n, N, C.c_float(thr_sq), _p(mask))
    return mask


def inlier_mask_pnp(P, X, Y, Z, u, v, thr_sq, N=None):
    P = _c(P, np.float32).reshape(12)
    a = [_c(w, np.float32) for w in (X, Y, Z, u, v)]
    n = a[0].size
    N = n if N is None else N
    mask = np.zeros((N + 63) // 64, np.uint64)
    lib().orc_inlier_mask_pnp(_p(P), *map(_p, a), n, N, C.c_float(thr_sq), _p(mask))
    return mask


def unpack_mask(words, N):
    bits = (np.asarray(words, np.uint64)[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)
    return bits.reshape(-1)[:N].astype(bool)


# ---------------------------------------------------------------- (3) triangulation
def triangulate(P1, P2, x1, y1, x2, y2, mask=None):
    P1, P2 = _c(P1, np.float32).reshape(12), _c(P2, np.float32).reshape(12)
    a = [_c(v, np.float32) for v in (x1, y1, x2, y2)]
    n = a[0].size
    XYZ = np.empty((n, 3), np.float32)
    m = None if mask is None else _c(mask, np.uint64)
    front = lib().orc_triangulate(_p(P1), _p(P2), *map(_p, a), n, _p(m), _p(XYZ))
    return XYZ, front


# ---------------------------------------------------------------- (4) projection / error (C)
def _d(a):
    return _c(a, np.float64)


def project_points(obj, rvec, tvec, K4, dist5):
    """validate_pose.py:181-184 — returns (uv float64 [n,2], uv_int int32 [n,2] truncated)."""
    obj = _c(obj, np.float32).reshape(-1, 3)
    n = obj.shape[0]
    uv = np.empty((n, 2), np.float64); uvi = np.empty((n, 2), np.int32)
    lib().orc_project_points(_p(obj), n, _p(_d(rvec).reshape(3)), _p(_d(tvec).reshape(3)), _p(_d(K4)), _p(_d(dist5)),
                             _p(uv), _p(uvi))
    return uv, uvi


def reprojection_error(obj, img, rvec, tvec, K4, dist5):
    """validate_pose.py:260-288 — returns (err [n], dict mean/std/max/min/num_points)."""
    obj = _c(obj, np.float32).reshape(-1, 3); img = _c(img, np.float32).reshape(-1, 2)
    n = obj.shape[0]
    err = np.empty(n, np.float64); st = np.empty(5, np.float64)
    lib().orc_reprojection_error(_p(obj), _p(img), n, _p(_d(rvec).reshape(3)), _p(_d(tvec).reshape(3)), _p(_d(K4)),
                                 _p(_d(dist5)), _p(err), _p(st))
    return err, {"mean": st[0], "std": st[1], "max": st[2], "min": st[3], "num_points": int(st[4])}


# ---------------------------------------------------------------- reference-level NumPy restatements
# constants: main.py:21-34 == validate_pose.py:24-37
MARKERS_X, MARKERS_Y = 11, 7
MARKER_LENGTH_MM, MARKER_SEPARATION_MM = 25.0, 10.0
CAMERA_MATRIX = np.array([[3108.95, 0, 1129.30], [0, 3101.48, 1940.89], [0, 0, 1]], dtype=np.float64)
DIST_COEFFS = np.array([0.148, -0.510, 0.003, -0.0004, 0.635], dtype=np.float64)
K4 = np.array([CAMERA_MATRIX[0, 0], CAMERA_MATRIX[1, 1], CAMERA_MATRIX[0, 2], CAMERA_MATRIX[1, 2]])


def board_object_points():
    """GridBoard((11,7), 0.025, 0.010) corner coordinates, float32 [77, 4, 3], id = j*11 + i.

    Geometry as the reference itself restates it at validate_pose.py:162-176 (marker (i, j) at
    x0 = i*0.035, y0 = j*0.035; corners (x0,y0),(x1,y0),(x1,y1),(x0,y1)); main.py:59-64 builds
    the same board through cv2.aruco.GridBoard.
    """
    m, s = MARKER_LENGTH_MM / 1000.0, MARKER_SEPARATION_MM / 1000.0
    pts = np.zeros((MARKERS_X * MARKERS_Y, 4, 3), np.float32)
    for j in range(MARKERS_Y):
        for i in range(MARKERS_X):
            x0, y0 = i * (m + s), j * (m + s)
            pts[j * MARKERS_X + i] = [[x0, y0, 0], [x0 + m, y0, 0], [x0 + m, y0 + m, 0], [x0, y0 + m, 0]]
    return pts


def rodrigues(rvec):
    """cv2.Rodrigues (main.py:122): R = cos I + (1-cos) k k' + sin [k]x."""
    r = np.asarray(rvec, np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3) + np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def rodrigues_inv(R):
    """Rotation matrix -> rotation vector (inverse of the above; angle in [0, pi])."""
    R = np.asarray(R, np.float64)
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    th = np.arctan2(0.5 * np.linalg.norm(w), c)      # not arccos(c): its slope is 1/sin(theta) next to 0 and pi
    if th < 1e-10:
        return w / 2
    if np.pi - th < 1e-6:
        A = ((R + R.T) / 2 - c * np.eye(3)) / (1 - c)      # = k k' exactly: sym(R) = cos I + (1 - cos) k k'
        k = np.sqrt(np.maximum(np.diag(A), 0))
        i = int(np.argmax(k))
        k = A[i] / k[i]
        if np.dot(w, k) < 0:
            k = -k
        return th * k
    return th * w / (2 * np.sin(th))


def quat_wxyz_from_matrix(R):
    """scipy Rotation.from_matrix(R).as_quat() reordered to [w,x,y,z] (main.py:138-141).

    Restated from SciPy's algorithm (largest of trace / diagonal picks the branch; no sign
    canonicalisation); pinned by SciPy-generated vectors in tests/golden/quaternions.json.
    """
    R = np.asarray(R, np.float64)
    d = [R[0, 0], R[1, 1], R[2, 2], R[0, 0] + R[1, 1] + R[2, 2]]
    c = int(np.argmax(d))
    q = np.empty(4)  # x, y, z, w
    if c != 3:
        i, j, k = c, (c + 1) % 3, (c + 2) % 3
        q[i] = 1 - d[3] + 2 * R[i, i]
        q[j] = R[j, i] + R[i, j]
        q[k] = R[k, i] + R[i, k]
        q[3] = R[k, j] - R[j, k]
    else:
        q[0] = R[2, 1] - R[1, 2]
        q[1] = R[0, 2] - R[2, 0]
        q[2] = R[1, 0] - R[0, 1]
        q[3] = 1 + d[3]
    q /= np.linalg.norm(q)
    return np.array([q[3], q[0], q[1], q[2]])


def camera_position_and_orientation(rvec, tvec):
    """get_camera_position_and_orientation, main.py:112-143 (forward = +Z per the code, not the comment)."""
    R = rodrigues(rvec)
    t = np.asarray(tvec, np.float64).reshape(3, 1)
    pos = -R.T @ t
    Rcw = R.T
    fwd = Rcw @ np.array([[0.0], [0.0], [1.0]])
    up = Rcw @ np.array([[0.0], [-1.0], [0.0]])
    return pos.flatten(), fwd.flatten(), up.flatten(), Rcw, quat_wxyz_from_matrix(Rcw)


def project_points_np(obj, rvec, tvec, K=CAMERA_MATRIX, dist=DIST_COEFFS):
    """NumPy twin of the C projection above (OpenCV-documented model)."""
    R = rodrigues(rvec)
    Xc = np.asarray(obj, np.float64).reshape(-1, 3) @ R.T + np.asarray(tvec, np.float64).reshape(1, 3)
    x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
    k1, k2, p1, p2, k3 = dist
    r2 = x * x + y * y
    g = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xd = x * g + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * g + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], axis=1)


def summary_of_means(means):
    """validate_pose.py:390-407: statistics of per-image means and the grade band."""
    m = float(np.mean(means))
    grade = "EXCELLENT" if m < 1.0 else "GOOD" if m < 2.0 else "FAIR" if m < 5.0 else "POOR"
    return {"mean": m, "max": float(np.max(means)), "min": float(np.min(means)), "std": float(np.std(means)),
            "grade": grade}


# ---------------------------------------------------------------- Levenberg-Marquardt pose refinement (NumPy checker)
def _lm_residual(p, obj, img, K, dist):
    uv = project_points_np(obj, p[:3], p[3:], K, dist)
    return (uv - img).reshape(-1)


def refine_pose_lm_np(obj, img, pose0, K=CAMERA_MATRIX, dist=DIST_COEFFS, max_iters=50):
    """Float64 LM on the distorted reprojection error with a central-difference Jacobian over (rvec, tvec):
    the independent checker of ph_refine_pose_lm (different parametrisation, numeric derivatives).
    Follows what cv2.solvePnP(SOLVEPNP_ITERATIVE) minimises (main.py:101-104).  Returns (pose [6], rms)."""
    obj = np.asarray(obj, np.float64).reshape(-1, 3); img = np.asarray(img, np.float64).reshape(-1, 2)
    pose = np.asarray(pose0, np.float64).copy()
    lam = 1e-3
    r = _lm_residual(pose, obj, img, K, dist); cost = r @ r

    def jac(p):
        J = np.empty((r.size, 6))
        for k in range(6):
            h = 1e-7 * max(1.0, abs(p[k])); d = np.zeros(6); d[k] = h
            J[:, k] = (_lm_residual(p + d, obj, img, K, dist) - _lm_residual(p - d, obj, img, K, dist)) / (2 * h)
        return J
    J = jac(pose)
    for _ in range(max_iters):
        JtJ, g = J.T @ J, J.T @ r
        step = np.linalg.solve(JtJ + lam * np.diag(np.diag(JtJ)), -g)
        r2 = _lm_residual(pose + step, obj, img, K, dist); c2 = r2 @ r2
        if c2 < cost:
            done = (cost - c2) <= 1e-14 * max(cost, 1e-300)
            pose, r, cost, lam = pose + step, r2, c2, max(lam / 10, 1e-12)
            J = jac(pose)
            if done:
                break
        else:
            lam *= 10
            if lam > 1e12:
                break
    return pose, float(np.sqrt(cost / len(obj)))
