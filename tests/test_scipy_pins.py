"""What the reference's own importable dependency (SciPy, /root/reference/main.py:138-141) can pin — CPU side.

`Rotation.from_rotvec(r).as_matrix()` IS cv2.Rodrigues (main.py:122) for every rotation vector: the goldens of
tests/golden/rodrigues.json (made by tests/golden/make_golden.py with SciPy) pin the oracle's and the host mirror's
Rodrigues in both directions, theta -> 0 and theta -> pi included, and the quaternion of the transposed matrix in
SciPy's own sign.  A second, independent Levenberg-Marquardt (scipy.optimize.least_squares on the a-5 projection
model) is run beside the oracle's NumPy LM for row a-3.

These pin Rodrigues and the quaternion ONLY: everything inside OpenCV (solvePnP's iteration schedule,
matchImagePoints' ordering, detectMarkers) stays "parity unpinned" (DESIGN.md §5).
"""
import json
import os

import numpy as np

from oracle import oracle as orc
from pose_estimation_amd import board, synthetic

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    d = json.load(open(os.path.join(G, "rodrigues.json")))
    assert "scipy" in d["source"] and len(d["cases"]) >= 60
    return [(np.array(c["rvec"]), np.array(c["R"]), np.array(c["rvec_back"]), np.array(c["quat_xyzw_of_Rt"])) for c in d["cases"]]


def branch_tie(M):
    """SciPy picks its quaternion branch by argmax(M00, M11, M22, trace); when the two largest agree to rounding (e.g. a
    rotation by exactly pi/2 about a coordinate axis: M11 = trace = 1) the branch — and with it the sign — is decided
    by the last bit of the matrix entries, so 'SciPy's sign' is not defined beyond 1e-15 perturbations of the input."""
    d = sorted([M[0, 0], M[1, 1], M[2, 2], M[0, 0] + M[1, 1] + M[2, 2]])
    return d[-1] - d[-2] < 1e-12


def test_rodrigues_forward_pinned_by_scipy():
    for r, R, _, _ in _cases():
        assert np.abs(orc.rodrigues(r) - R).max() < 1e-14, r
        assert np.abs(board.rodrigues(r) - R).max() < 1e-14, r


def test_rodrigues_inverse_pinned_by_scipy():
    for r, R, back, _ in _cases():
        th = np.linalg.norm(back)
        # next to pi the angle comes from arccos((tr - 1) / 2) (slope 1 / sin theta) and the axis from sqrt of 1 + diagonal
        # entries: digits are lost in R itself (|dR| ~ 1e-16 -> |dr| up to ~1e-8); SciPy goes through the quaternion
        tol = min(2e-7, max(1e-12, 1e-13 / (np.pi - th) ** 2)) if th < np.pi else 2e-7
        for f in (orc.rodrigues_inv, board.rotation_to_rvec):
            got = f(R)
            assert np.abs(got - back).max() < tol, (r, got, back)
            assert np.abs(orc.rodrigues(got) - R).max() < 1e-9


def test_quaternion_of_transpose_pinned_by_scipy_with_sign():
    """main.py:138-141 on R_cam_to_world = R.T: SciPy's sign, not 'up to sign'."""
    ties = 0
    for r, R, _, q in _cases():
        want = np.array([q[3], q[0], q[1], q[2]])
        assert np.abs(orc.quat_wxyz_from_matrix(R.T) - want).max() < 1e-12, r
        assert np.abs(board._quat_xyzw(R.T) - q).max() < 1e-12, r
        _, _, _, Rcw, quat = board.get_camera_position_and_orientation(r, np.zeros(3))
        assert np.abs(Rcw - R.T).max() < 1e-14
        if branch_tie(R.T):             # SciPy's branch (hence sign) hangs on the last bit of the matrix: documented tie
            ties += 1
            assert min(np.abs(quat - want).max(), np.abs(quat + want).max()) < 1e-12, r
        else:
            assert np.abs(quat - want).max() < 1e-12, r
    assert ties <= 2


def test_numpy_lm_agrees_with_scipy_least_squares():
    """Row a-3, second independent LM: scipy.optimize.least_squares (method 'lm', MINPACK) on the same distorted
    reprojection residuals, from the same perturbed start, lands where the oracle's NumPy LM lands."""
    from scipy.optimize import least_squares
    views = synthetic.board_views(3, seed=21, noise_px=0.3)
    obj_all = orc.board_object_points()
    rng = np.random.default_rng(4)
    for v in views:
        ids = v["ids"].flatten()
        obj = obj_all[ids].reshape(-1, 3).astype(np.float64)
        img = np.concatenate([c.reshape(4, 2) for c in v["corners"]]).astype(np.float64)
        p0 = np.concatenate([orc.rodrigues_inv(v["R"]), v["t"]])
        p0[:3] += rng.normal(0, 0.03, 3); p0[3:] += rng.normal(0, 0.01, 3)

        def resid(p):
            return (orc.project_points_np(obj, p[:3], p[3:]) - img).ravel()

        sp = least_squares(resid, p0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15).x
        mine, rms = orc.refine_pose_lm_np(obj, img, p0, max_iters=100)
        assert np.abs(orc.rodrigues(sp[:3]) - orc.rodrigues(mine[:3])).max() < 1e-7
        assert np.abs(sp[3:] - mine[3:]).max() < 1e-7
        assert abs(np.sqrt(np.mean(resid(sp) ** 2) * 2) - rms) < 1e-7        # rms over points (two residuals each)
