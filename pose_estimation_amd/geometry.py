"""Host-side (NumPy, float64) pieces that north_star leaves on the host: the recoverPose-style
decomposition of an essential matrix into its four (R, t) candidates and small pose utilities.
There is no reference code for these (main.py only ever calls cv2.solvePnP); the candidate
construction is the textbook one (Hartley & Zisserman, Result 9.19)."""
from __future__ import annotations

import numpy as np

_W = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])


def decompose_essential(E):
    """E [..., 3, 3] -> (R [..., 4, 3, 3], t [..., 4, 3]): (R1,t), (R1,-t), (R2,t), (R2,-t)."""
    E = np.asarray(E, np.float64)
    U, _, Vt = np.linalg.svd(E)
    U = U * np.where(np.linalg.det(U) < 0, -1.0, 1.0)[..., None, None]
    Vt = Vt * np.where(np.linalg.det(Vt) < 0, -1.0, 1.0)[..., None, None]
    R1 = U @ _W @ Vt
    R2 = U @ _W.T @ Vt
    t = U[..., :, 2]
    R = np.stack([R1, R1, R2, R2], axis=-3)
    tt = np.stack([t, -t, t, -t], axis=-2)
    return R, tt


def candidate_projections(E):
    """E [B, 3, 3] -> (P1 [B, 4, 12], P2 [B, 4, 12]) float32, normalised coordinates, P1 = [I|0]."""
    R, t = decompose_essential(E)
    B = R.shape[0]
    P2 = np.concatenate([R, t[..., None]], axis=-1).reshape(B, 4, 12)
    P1 = np.broadcast_to(np.concatenate([np.eye(3), np.zeros((3, 1))], axis=1).reshape(1, 1, 12), (B, 4, 12))
    return np.ascontiguousarray(P1, np.float32), np.ascontiguousarray(P2, np.float32), R, t


def rotation_angle_deg(Ra, Rb):
    """Angle of Ra Rb' in degrees, as atan2(|antisymmetric part|, (trace - 1) / 2): arccos of the trace alone resolves
    nothing below sqrt(eps) (0.02 degrees for matrices stored as float32)."""
    D = np.asarray(Ra, np.float64) @ np.asarray(Rb, np.float64).T
    s = 0.5 * np.sqrt((D[2, 1] - D[1, 2]) ** 2 + (D[0, 2] - D[2, 0]) ** 2 + (D[1, 0] - D[0, 1]) ** 2)
    return float(np.degrees(np.arctan2(s, (np.trace(D) - 1) / 2)))


def direction_angle_deg(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.degrees(np.arctan2(np.linalg.norm(np.cross(a, b)), np.dot(a, b))))
