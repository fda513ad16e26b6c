"""Board-pose functions of the reference, same names and argument meaning, on the MI355X path.

Mirrors (reference file:line):
  constants                              main.py:20-41 == validate_pose.py:23-46
  create_aruco_board                     main.py:56-65           (geometry only; the ArUco
                                         dictionary bit patterns live inside OpenCV)
  estimate_board_pose                    main.py:86-109 == validate_pose.py:83-101
  get_camera_position_and_orientation    main.py:112-143
  project_points                         validate_pose.py:181-184
  calculate_reprojection_error           validate_pose.py:260-288
  create_board_3d_points                 validate_pose.py:104-178

Error convention as in the reference: sentinel returns (None, None, n), never exceptions, for
"no markers / too few points / solver failed" (main.py:91-107).

Where the arithmetic runs: everything is a batched float64 HIP kernel (csrc/board.hip) behind the C ABI.  The pose
solve follows cv2.solvePnP(SOLVEPNP_ITERATIVE) on coplanar points as documented: the homography-based initial pose
(`ph_initial_board_pose`, one block per image: normalised DLT normal matrix, Jacobi eigenvector, Newton polar factor),
then Levenberg-Marquardt on the distorted reprojection error (`ph_refine_pose_lm`, one block per image) — two launches
for all images of a folder, no host linear algebra.  Projection, reprojection-error statistics and the camera
inversion / quaternion epilogue (`ph_camera_from_pose`) are one launch each.  cv2 is not installed here, so the solve is
"parity unpinned" against OpenCV's exact iteration schedule.  The NumPy twins kept in this file (`rodrigues`,
`_homography_pose`, the single-pose `get_camera_position_and_orientation`) serve callers without a batch and the CPU tests.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L

# ----------------------------------------------------------------------------- configuration
MARKERS_X = 11
MARKERS_Y = 7
MARKER_LENGTH_MM = 25.0
MARKER_SEPARATION_MM = 10.0
ARUCO_DICT = "DICT_6X6_250"  # cv2.aruco.DICT_6X6_250 in the reference; only the name is needed here

CAMERA_MATRIX = np.array([
    [3108.95, 0, 1129.30],
    [0, 3101.48, 1940.89],
    [0, 0, 1]
], dtype=np.float64)

DIST_COEFFS = np.array([0.148, -0.510, 0.003, -0.0004, 0.635], dtype=np.float64)

ROTATE_LANDSCAPE_TO_PORTRAIT = True
IMAGE_FOLDER = "test_data/ChArUco"
OUTPUT_FOLDER = "test_data/pose_validation"


class GridBoard:
    """Geometry of cv2.aruco.GridBoard((mx, my), marker_length, separation, dictionary).

    Marker (i, j) has id j*mx + i and corners (x0,y0),(x0+l,y0),(x0+l,y0+l),(x0,y0+l) with
    x0 = i*(l+s), y0 = j*(l+s), z = 0 — the geometry the reference itself restates at
    validate_pose.py:162-176.
    """

    def __init__(self, size, marker_length, marker_separation, dictionary=ARUCO_DICT):
        self.size = tuple(size)
        self.marker_length = float(marker_length)
        self.marker_separation = float(marker_separation)
        self.dictionary = dictionary
        mx, my = self.size
        l, s = self.marker_length, self.marker_separation
        pts = np.zeros((mx * my, 4, 3), np.float32)
        for j in range(my):
            for i in range(mx):
                x0, y0 = i * (l + s), j * (l + s)
                pts[j * mx + i] = [[x0, y0, 0], [x0 + l, y0, 0], [x0 + l, y0 + l, 0], [x0, y0 + l, 0]]
        self.obj_points = pts
        self.ids = np.arange(mx * my, dtype=np.int32)

    def getObjPoints(self):
        return self.obj_points

    def getIds(self):
        return self.ids

    def matchImagePoints(self, corners, ids):
        """Object/image point pairs of the detected markers that belong to the board.

        Shapes as OpenCV returns them: (4n,1,3) float32 and (4n,1,2) float32; (None, None)
        when nothing matches.
        """
        if ids is None or len(ids) == 0:
            return None, None
        obj, img = [], []
        for c, m in zip(corners, np.asarray(ids).reshape(-1)):
            m = int(m)
            if 0 <= m < len(self.obj_points):
                obj.append(self.obj_points[m])
                img.append(np.asarray(c, np.float32).reshape(4, 2))
        if not obj:
            return None, None
        return (np.concatenate(obj).reshape(-1, 1, 3).astype(np.float32),
                np.concatenate(img).reshape(-1, 1, 2).astype(np.float32))


def create_aruco_board():
    """main.py:56-65 — returns (dictionary, board)."""
    dictionary = ARUCO_DICT
    board = GridBoard((MARKERS_X, MARKERS_Y), MARKER_LENGTH_MM / 1000.0, MARKER_SEPARATION_MM / 1000.0, dictionary)
    return dictionary, board


# ----------------------------------------------------------------------------- device plumbing
def _device():
    if not torch.cuda.is_available():
        raise L.PosehotError("no GPU visible: the pose path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _k4(camera_matrix):
    K = np.asarray(camera_matrix, np.float64)
    return np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], np.float64)


def _cam_tensors(camera_matrix, dist_coeffs, dev):
    return (torch.from_numpy(_k4(camera_matrix)).to(dev),
            torch.from_numpy(np.asarray(dist_coeffs, np.float64).reshape(-1)[:5].copy()).to(dev))


def project_points_batched(obj, pose, camera_matrix=CAMERA_MATRIX, dist_coeffs=DIST_COEFFS, n_valid=None):
    """obj [B,N,3] f32, pose [B,6] f64 (rvec,tvec) on the GPU -> (uv f64 [B,N,2], uv_int int32 [B,N,2])."""
    lib = L.load()
    obj, pose = L.dev(obj, torch.float32, "obj"), L.dev(pose, torch.float64, "pose")
    B, N = obj.shape[0], obj.shape[1]
    K4, d5 = _cam_tensors(camera_matrix, dist_coeffs, obj.device)
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    uv = torch.zeros((B, N, 2), dtype=torch.float64, device=obj.device)
    uvi = torch.zeros((B, N, 2), dtype=torch.int32, device=obj.device)
    L.check(lib.ph_project_points(L.ptr(obj), L.ptr(pose), L.ptr(K4), L.ptr(d5), L.ptr(n_valid), B, N,
                                  L.ptr(uv), L.ptr(uvi), L.stream_ptr()), "ph_project_points")
    return uv, uvi


def reprojection_error_batched(obj, img, pose, n_valid=None, camera_matrix=CAMERA_MATRIX, dist_coeffs=DIST_COEFFS):
    """-> (err f64 [B,N], stats f64 [B,5] = mean, std, max, min, num_points)."""
    lib = L.load()
    obj, img = L.dev(obj, torch.float32, "obj"), L.dev(img, torch.float32, "img")
    pose = L.dev(pose, torch.float64, "pose")
    B, N = obj.shape[0], obj.shape[1]
    K4, d5 = _cam_tensors(camera_matrix, dist_coeffs, obj.device)
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    err = torch.zeros((B, N), dtype=torch.float64, device=obj.device)
    stats = torch.zeros((B, 5), dtype=torch.float64, device=obj.device)
    L.check(lib.ph_reprojection_error(L.ptr(obj), L.ptr(img), L.ptr(pose), L.ptr(K4), L.ptr(d5), L.ptr(n_valid),
                                      B, N, L.ptr(err), L.ptr(stats), L.stream_ptr()), "ph_reprojection_error")
    return err, stats


def undistort_points(uv, camera_matrix=CAMERA_MATRIX, dist_coeffs=DIST_COEFFS, iters=5):
    """uv [n,2] f32 pixels on the GPU -> normalised undistorted [n,2] f32."""
    lib = L.load()
    uv = L.dev(uv, torch.float32, "uv")
    K4, d5 = _cam_tensors(camera_matrix, dist_coeffs, uv.device)
    out = torch.empty_like(uv)
    L.check(lib.ph_undistort_points(L.ptr(uv), L.ptr(K4), L.ptr(d5), uv.shape[0], int(iters), L.ptr(out),
                                    L.stream_ptr()), "ph_undistort_points")
    return out


# ----------------------------------------------------------------------------- reference functions
def project_points(points_3d, rvec, tvec, camera_matrix, dist_coeffs):
    """validate_pose.py:181-184 — (M,3) -> (M,2) int32, truncated toward zero."""
    dev = _device()
    obj = torch.from_numpy(np.asarray(points_3d, np.float32).reshape(1, -1, 3).copy()).to(dev)
    pose = torch.from_numpy(np.concatenate([np.asarray(rvec, np.float64).reshape(3),
                                            np.asarray(tvec, np.float64).reshape(3)])[None]).to(dev)
    _, uvi = project_points_batched(obj, pose, camera_matrix, dist_coeffs)
    return uvi[0].cpu().numpy().reshape(-1, 2).astype(np.int32)


def calculate_reprojection_error(detected_corners, detected_ids, board, rvec, tvec, camera_matrix, dist_coeffs):
    """validate_pose.py:260-288 — dict mean/std/max/min/num_points, or None."""
    if detected_ids is None or len(detected_ids) == 0:
        return None
    obj_points, img_points = board.matchImagePoints(detected_corners, detected_ids)
    if obj_points is None:
        return None
    dev = _device()
    obj = torch.from_numpy(obj_points.reshape(1, -1, 3)).to(dev)
    img = torch.from_numpy(img_points.reshape(1, -1, 2)).to(dev)
    pose = torch.from_numpy(np.concatenate([np.asarray(rvec, np.float64).reshape(3),
                                            np.asarray(tvec, np.float64).reshape(3)])[None]).to(dev)
    _, stats = reprojection_error_batched(obj, img, pose, None, camera_matrix, dist_coeffs)
    s = stats[0].cpu().numpy()
    return {"mean": s[0], "std": s[1], "max": s[2], "min": s[3], "num_points": int(s[4])}


def rodrigues(rvec):
    """cv2.Rodrigues(rvec)[0] (main.py:122)."""
    r = np.asarray(rvec, np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3) + np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def rotation_to_rvec(R):
    R = np.asarray(R, np.float64)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    th = np.arctan2(0.5 * np.linalg.norm(w), np.clip((np.trace(R) - 1) / 2, -1, 1))   # accurate next to 0 and pi, unlike arccos
    if th < 1e-10:
        return w / 2
    if np.pi - th < 1e-6:
        c = np.clip((np.trace(R) - 1) / 2, -1, 1)
        A = ((R + R.T) / 2 - c * np.eye(3)) / (1 - c)      # = k k' exactly: sym(R) = cos I + (1 - cos) k k'
        d = np.sqrt(np.maximum(np.diag(A), 0))
        i = int(np.argmax(d))
        k = A[i] / d[i]
        return th * (k if np.dot(w, k) >= 0 else -k)
    return th * w / (2 * np.sin(th))


def _quat_xyzw(Rm):
    """Rotation.from_matrix(Rm).as_quat() of SciPy for a proper rotation (no canonicalisation)."""
    d = [Rm[0, 0], Rm[1, 1], Rm[2, 2], Rm[0, 0] + Rm[1, 1] + Rm[2, 2]]
    c = int(np.argmax(d))
    q = np.empty(4)
    if c != 3:
        i, j, k = c, (c + 1) % 3, (c + 2) % 3
        q[i] = 1 - d[3] + 2 * Rm[i, i]
        q[j] = Rm[j, i] + Rm[i, j]
        q[k] = Rm[k, i] + Rm[i, k]
        q[3] = Rm[k, j] - Rm[j, k]
    else:
        q[0] = Rm[2, 1] - Rm[1, 2]
        q[1] = Rm[0, 2] - Rm[2, 0]
        q[2] = Rm[1, 0] - Rm[0, 1]
        q[3] = 1 + d[3]
    return q / np.linalg.norm(q)


def get_camera_position_and_orientation(rvec, tvec):
    """main.py:112-143 — (position, forward, up, R_cam_to_world, quaternion [w,x,y,z]) of ONE pose, on the host.

    The pose maps board to camera coordinates, x_cam = R x_board + t, so the camera sits at -R't and its axes, seen from
    the board, are the columns of R': forward is the camera's +Z (what the reference's code does, main.py:132, not what its
    comment says), up its -Y (main.py:135).  The quaternion is SciPy's ``Rotation.from_matrix(R').as_quat()`` — sign included,
    pinned by tests/golden/quaternions.json — reordered to w first (main.py:138-141).  Batches go through
    `camera_from_pose_batched` (one `ph_camera_from_pose` launch); the two agree to 1e-14 (tests/test_gpu_pnp.py)."""
    Rcw = rodrigues(rvec).T
    centre = -(Rcw @ np.asarray(tvec, np.float64).reshape(3))
    x, y, z, w = _quat_xyzw(Rcw)
    return centre, Rcw[:, 2].copy(), -Rcw[:, 1], Rcw, np.array([w, x, y, z])


def camera_from_pose_batched(pose):
    """get_camera_position_and_orientation (main.py:112-143) for B poses in one launch.

    pose [B,6] f64 on the GPU -> dict of device tensors: position [B,3], forward [B,3], up [B,3],
    R_cam_to_world [B,3,3], quaternion [B,4] in [w,x,y,z] order."""
    lib = L.load()
    pose = L.dev(pose, torch.float64, "pose")
    B = pose.shape[0]
    out = torch.empty((B, 22), dtype=torch.float64, device=pose.device)
    L.check(lib.ph_camera_from_pose(L.ptr(pose), B, L.ptr(out), L.stream_ptr()), "ph_camera_from_pose")
    return dict(position=out[:, 0:3], forward=out[:, 3:6], up=out[:, 6:9], R_cam_to_world=out[:, 9:18].reshape(B, 3, 3),
                quaternion=out[:, 18:22])


# ----------------------------------------------------------------------------- solvePnP(ITERATIVE), planar
def _undistort_np(uv, K, dist, iters=20):
    k1, k2, p1, p2, k3 = dist
    x0 = (uv[:, 0] - K[0, 2]) / K[0, 0]
    y0 = (uv[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icd = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * icd, (y0 - dy) * icd
    return np.stack([x, y], axis=1)


def _homography_pose(obj, xy):
    """Initial pose of a z = 0 point set from the plane-to-image homography (normalised DLT)."""
    P = obj[:, :2]
    mo, mi = P.mean(0), xy.mean(0)
    so = np.sqrt(2) / max(np.sqrt(((P - mo) ** 2).sum(1)).mean(), 1e-12)
    si = np.sqrt(2) / max(np.sqrt(((xy - mi) ** 2).sum(1)).mean(), 1e-12)
    To = np.array([[so, 0, -so * mo[0]], [0, so, -so * mo[1]], [0, 0, 1]])
    Ti = np.array([[si, 0, -si * mi[0]], [0, si, -si * mi[1]], [0, 0, 1]])
    Pn = (P - mo) * so
    xn = (xy - mi) * si
    n = len(P)
    A = np.zeros((2 * n, 9))
    A[0::2, 0:2], A[0::2, 2], A[0::2, 6:8], A[0::2, 8] = Pn, 1, -xn[:, :1] * Pn, -xn[:, 0]
    A[1::2, 3:5], A[1::2, 5], A[1::2, 6:8], A[1::2, 8] = Pn, 1, -xn[:, 1:] * Pn, -xn[:, 1]
    _, _, Vt = np.linalg.svd(A)
    Hm = np.linalg.inv(Ti) @ Vt[-1].reshape(3, 3) @ To
    h1, h2, h3 = Hm[:, 0], Hm[:, 1], Hm[:, 2]
    lam = 2.0 / (np.linalg.norm(h1) + np.linalg.norm(h2))
    if h3[2] * lam < 0:  # board in front of the camera
        lam = -lam
    r1, r2, t = lam * h1, lam * h2, lam * h3
    U, _, Vt2 = np.linalg.svd(np.stack([r1, r2, np.cross(r1, r2)], axis=1))
    R = U @ Vt2
    if np.linalg.det(R) < 0:
        R = U @ np.diag([1, 1, -1]) @ Vt2
    return R, t


def refine_pose_lm_batched(obj, img, pose, n_valid=None, camera_matrix=CAMERA_MATRIX, dist_coeffs=DIST_COEFFS,
                           max_iters=30):
    """Batched Levenberg-Marquardt on the distorted reprojection error (device, float64).

    obj [B,N,3] f32, img [B,N,2] f32, pose [B,6] f64 initial (rvec, tvec) — all on the GPU.
    Returns (pose [B,6] refined, rms [B], iters [B] int32)."""
    lib = L.load()
    obj, img = L.dev(obj, torch.float32, "obj"), L.dev(img, torch.float32, "img")
    pose = L.dev(pose, torch.float64, "pose").clone()
    B, N = obj.shape[0], obj.shape[1]
    K4, d5 = _cam_tensors(camera_matrix, dist_coeffs, obj.device)
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    rms = torch.zeros((B,), dtype=torch.float64, device=obj.device)
    iters = torch.zeros((B,), dtype=torch.int32, device=obj.device)
    L.check(lib.ph_refine_pose_lm(L.ptr(obj), L.ptr(img), L.ptr(n_valid), B, N, L.ptr(K4), L.ptr(d5), int(max_iters),
                                  L.ptr(pose), L.ptr(rms), L.ptr(iters), L.stream_ptr()), "ph_refine_pose_lm")
    return pose, rms, iters


def initial_board_pose(obj_points, img_points, camera_matrix, dist_coeffs):
    """Homography-based initial pose of a coplanar (z = 0) point set: (rvec, tvec) float64 [6], or None."""
    obj = np.asarray(obj_points, np.float64).reshape(-1, 3)
    img = np.asarray(img_points, np.float64).reshape(-1, 2)
    K = np.asarray(camera_matrix, np.float64)
    dist = np.asarray(dist_coeffs, np.float64).reshape(-1)[:5]
    if len(obj) < 4 or len(obj) != len(img):
        return None
    try:
        R0, t0 = _homography_pose(obj, _undistort_np(img, K, dist))
    except np.linalg.LinAlgError:
        return None
    pose = np.concatenate([rotation_to_rvec(R0), t0])
    return pose if np.all(np.isfinite(pose)) else None


def initial_board_pose_batched(obj, img, n_valid=None, camera_matrix=CAMERA_MATRIX, dist_coeffs=DIST_COEFFS):
    """Homography-based initial poses of many coplanar point sets on the device (``ph_initial_board_pose``).

    obj [B,N,3] f32, img [B,N,2] f32 on the GPU.  Returns (pose [B,6] f64, ok [B] int32)."""
    lib = L.load()
    obj, img = L.dev(obj, torch.float32, "obj"), L.dev(img, torch.float32, "img")
    B, N = obj.shape[0], obj.shape[1]
    K4, d5 = _cam_tensors(camera_matrix, dist_coeffs, obj.device)
    n_valid = L.opt_i32(n_valid, B, "n_valid")
    pose = torch.empty((B, 6), dtype=torch.float64, device=obj.device)
    ok = torch.empty((B,), dtype=torch.int32, device=obj.device)
    L.check(lib.ph_initial_board_pose(L.ptr(obj), L.ptr(img), L.ptr(n_valid), B, N, L.ptr(K4), L.ptr(d5), L.ptr(pose),
                                      L.ptr(ok), L.stream_ptr()), "ph_initial_board_pose")
    return pose, ok


def solve_pnp_iterative_batched(items, camera_matrix, dist_coeffs, max_iters=30, init="device"):
    """cv2.solvePnP(..., SOLVEPNP_ITERATIVE) for many coplanar point sets in two launches.

    items: list of (obj_points, img_points).  Returns a list of (success, rvec (3,1), tvec (3,1)):
    homography initial pose (init="device": one ``ph_initial_board_pose`` launch for all items; "host": a NumPy SVD per
    item, the twin the device kernel is tested against), then ONE batched Levenberg-Marquardt launch on the GPU
    (``ph_refine_pose_lm``)."""
    if init not in ("device", "host"):
        raise ValueError("init must be 'device' or 'host'")
    out = [(False, None, None)] * len(items)
    shaped = []
    for o, i in items:
        o = np.asarray(o, np.float32).reshape(-1, 3); i = np.asarray(i, np.float32).reshape(-1, 2)
        shaped.append((o, i) if len(o) >= 4 and len(o) == len(i) else None)
    if init == "host":
        inits = [None if s is None else initial_board_pose(s[0], s[1], camera_matrix, dist_coeffs) for s in shaped]
        good = [k for k, p in enumerate(inits) if p is not None]
    else:
        good = [k for k, s in enumerate(shaped) if s is not None]
    if not good:
        return out
    dev = _device()
    N = max(len(shaped[k][0]) for k in good)
    obj = np.zeros((len(good), N, 3), np.float32); img = np.zeros((len(good), N, 2), np.float32)
    nv = np.zeros(len(good), np.int32); pose = np.zeros((len(good), 6), np.float64)
    for r, k in enumerate(good):
        o, i = shaped[k]
        nv[r] = len(o); obj[r, :len(o)] = o; img[r, :len(o)] = i
        if init == "host":
            pose[r] = inits[k]
    t = lambda a: torch.from_numpy(a).to(dev)
    obj_d, img_d, nv_d = t(obj), t(img), t(nv)
    if init == "device":
        pose_d, ok0 = initial_board_pose_batched(obj_d, img_d, nv_d, camera_matrix, dist_coeffs)
        pose_d = torch.where(ok0[:, None] != 0, pose_d, torch.zeros_like(pose_d))     # a failed start stays out of the LM's way
        ok0 = ok0.cpu().numpy()
    else:
        pose_d, ok0 = t(pose), np.ones(len(good), np.int32)
    refined, _, _ = refine_pose_lm_batched(obj_d, img_d, pose_d, nv_d, camera_matrix, dist_coeffs, max_iters)
    refined = refined.cpu().numpy()
    for r, k in enumerate(good):
        ok = bool(ok0[r]) and bool(np.all(np.isfinite(refined[r])))
        out[k] = (ok, refined[r, :3].reshape(3, 1).copy(), refined[r, 3:].reshape(3, 1).copy()) if ok else (False, None, None)
    return out


def solve_pnp_iterative(obj_points, img_points, camera_matrix, dist_coeffs, max_iters=30):
    """cv2.solvePnP(..., flags=SOLVEPNP_ITERATIVE) for one coplanar point set: (success, rvec, tvec)."""
    return solve_pnp_iterative_batched([(obj_points, img_points)], camera_matrix, dist_coeffs, max_iters)[0]


def estimate_board_pose(corners, ids, board, camera_matrix, dist_coeffs):
    """main.py:86-109 — (rvec, tvec, num_markers) or (None, None, n) sentinels."""
    if ids is None or len(ids) == 0:
        return None, None, 0
    obj_points, img_points = board.matchImagePoints(corners, ids)
    if obj_points is None or len(obj_points) < 4:
        return None, None, len(ids) if ids is not None else 0
    success, rvec, tvec = solve_pnp_iterative(obj_points, img_points, camera_matrix, dist_coeffs)
    if not success:
        return None, None, len(ids)
    return rvec, tvec, len(ids)


def estimate_board_poses(detections, board, camera_matrix, dist_coeffs):
    """estimate_board_pose for a list of (corners, ids) with ONE refinement launch for all of them.

    Same guards and sentinels per item as main.py:86-109: returns [(rvec, tvec, num_markers), ...]."""
    out, items, where = [], [], []
    for k, (corners, ids) in enumerate(detections):
        if ids is None or len(ids) == 0:
            out.append((None, None, 0))
            continue
        out.append((None, None, len(ids)))
        obj_points, img_points = board.matchImagePoints(corners, ids)
        if obj_points is None or len(obj_points) < 4:
            continue
        items.append((obj_points, img_points))
        where.append(k)
    if items:
        for k, (success, rvec, tvec) in zip(where, solve_pnp_iterative_batched(items, camera_matrix, dist_coeffs)):
            if success:
                out[k] = (rvec, tvec, out[k][2])
    return out


def create_board_3d_points(board_width_m, board_height_m):
    """validate_pose.py:104-178 — point sets of the validation overlay (float32 where the reference's are)."""
    points = {}
    points["boundary"] = np.array([[0, 0, 0], [board_width_m, 0, 0], [board_width_m, board_height_m, 0],
                                   [0, board_height_m, 0], [0, 0, 0]], dtype=np.float32)
    marker_size_m = MARKER_LENGTH_MM / 1000.0
    sep_m = MARKER_SEPARATION_MM / 1000.0
    vertical_lines = []
    for i in range(MARKERS_X + 1):   # only the last if/elif/else of validate_pose.py:131-136 takes effect
        x = 0 if i == 0 else board_width_m if i == MARKERS_X else i * (marker_size_m + sep_m)
        vertical_lines.append([[x, 0, 0], [x, board_height_m, 0]])
    points["vertical_lines"] = vertical_lines
    horizontal_lines = []
    for j in range(MARKERS_Y + 1):
        y = 0 if j == 0 else board_height_m if j == MARKERS_Y else j * (marker_size_m + sep_m)
        horizontal_lines.append([[0, y, 0], [board_width_m, y, 0]])
    points["horizontal_lines"] = horizontal_lines
    centers, boxes = [], []
    for j in range(MARKERS_Y):
        for i in range(MARKERS_X):
            x0, y0 = i * (marker_size_m + sep_m), j * (marker_size_m + sep_m)
            x1, y1 = x0 + marker_size_m, y0 + marker_size_m
            centers.append([x0 + marker_size_m / 2, y0 + marker_size_m / 2, 0])
            boxes.append(np.array([[x0, y0, 0], [x1, y0, 0], [x1, y1, 0], [x0, y1, 0], [x0, y0, 0]], dtype=np.float32))
    points["marker_centers"] = np.array(centers, dtype=np.float32)
    points["marker_boxes"] = boxes
    return points
