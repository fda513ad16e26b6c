"""Generates the committed fixtures of tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference holds no fixtures, tests or data, and cv2 is not installed, so the
only vectors that come from the reference's own dependency set are the SciPy quaternions
(main.py:138-141 calls scipy Rotation.from_matrix(...).as_quat()).  Everything else is a small
seeded input with the CPU oracle's output, frozen so that a change of the oracle or of the
kernels' arithmetic contract shows up as a diff against history.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as orc  # noqa: E402
from pose_estimation_amd import synthetic  # noqa: E402


def quaternions():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(42)
    cases = []
    mats = [Rotation.from_rotvec(rng.normal(0, 1.5, 3)).as_matrix() for _ in range(40)]
    mats += [np.eye(3), np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0]), np.diag([-1.0, -1.0, 1.0])]
    mats += [Rotation.from_rotvec([np.pi - 1e-4, 0, 0]).as_matrix(), Rotation.from_rotvec([0, 0, 3.0]).as_matrix()]
    for M in mats:
        q = Rotation.from_matrix(M).as_quat()      # [x, y, z, w]
        cases.append({"R": M.tolist(), "quat_xyzw": q.tolist()})
    json.dump({"source": "scipy " + __import__("scipy").__version__ + " Rotation.from_matrix(R).as_quat()", "cases": cases},
              open(os.path.join(HERE, "quaternions.json"), "w"), indent=0)


def rodrigues():
    """cv2.Rodrigues (main.py:122) == scipy Rotation.from_rotvec(r).as_matrix(): the one piece of the OpenCV side
    that the reference's own dependency set can pin.  Stored per case: the rotation vector, SciPy's matrix, SciPy's
    rotation vector back from that matrix (the inverse direction) and main.py:138-141's quaternion of the TRANSPOSE
    (the camera-to-world rotation the reference converts), in SciPy's own sign."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(2024)
    vecs = [rng.normal(0, 1.2, 3) for _ in range(40)]
    vecs += [np.zeros(3), np.array([1e-13, -2e-13, 3e-14]), np.array([1e-9, 0, 0]), np.array([0, -3e-7, 1e-7]),
             np.array([1e-4, 2e-4, -1e-4])]                                                  # theta -> 0
    for th in (np.pi - 1e-3, np.pi - 1e-5, np.pi - 1e-7, np.pi - 1e-9):                      # theta -> pi
        for ax in (np.array([1.0, 0, 0]), np.array([0.6, -0.48, 0.64]), np.array([0.0, 0.6, 0.8]), np.array([-0.36, 0.48, -0.8])):
            vecs.append(th * ax / np.linalg.norm(ax))
    vecs += [np.array([0.0, 0.0, 3.0]), np.array([2.0, -2.0, 0.5]), np.array([0.0, np.pi / 2, 0.0])]
    cases = []
    for r in vecs:
        rot = Rotation.from_rotvec(r)
        M = rot.as_matrix()
        cases.append({"rvec": [float(x) for x in r], "R": M.tolist(), "rvec_back": Rotation.from_matrix(M).as_rotvec().tolist(),
                      "quat_xyzw_of_Rt": Rotation.from_matrix(M.T).as_quat().tolist()})
    json.dump({"source": "scipy " + __import__("scipy").__version__ + " Rotation.from_rotvec(r).as_matrix() / from_matrix(R).as_rotvec() / "
                         "from_matrix(R.T).as_quat()", "cases": cases},
              open(os.path.join(HERE, "rodrigues.json"), "w"), indent=0)


def matching():
    rng = np.random.default_rng(7)
    q = rng.integers(0, 256, (96, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (160, 32), dtype=np.uint8)
    t[150] = t[4]; q[0] = t[4]; q[1] = t[150] ^ np.uint8(1)
    hi, h1, h2 = orc.match_hamming_top2(q, t)
    fq = synthetic.sift_like(rng, 64, 128); ft = synthetic.sift_like(rng, 96, 128)
    ft[90] = ft[3]; fq[0] = ft[3]
    li, l1, l2 = orc.match_l2_top2(fq, ft)
    np.savez_compressed(os.path.join(HERE, "matching.npz"), hq=q, ht=t, h_idx=hi, h_d1=h1, h_d2=h2,
                        fq=fq, ft=ft, l_idx=li, l_d1sq=l1, l_d2sq=l2)


def scoring():
    rng = np.random.default_rng(11)
    n, H = 200, 12
    R = synthetic.rot_y(np.deg2rad(-10.0)); t = -R @ np.array([0.5, 0.0, 0.0])
    X = rng.uniform(-1, 1, (n, 3)) + [0, 0, 4.0]
    x1 = (X[:, :2] / X[:, 2:3]).astype(np.float32)
    Xc = X @ R.T + t
    x2 = (Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, 1 / synthetic.FX, (n, 2))).astype(np.float32)
    x2[::5] = rng.uniform(-0.3, 0.3, (len(x2[::5]), 2)).astype(np.float32)
    E = np.stack([synthetic.essential_from_pose((np.eye(3) + synthetic.skew(rng.normal(0, 0.01 * (h > 0), 3))) @ R,
                                                t + rng.normal(0, 0.02 * (h > 0), 3)).reshape(9) for h in range(H)]).astype(np.float32)
    thr = np.float32((1.5 / synthetic.FX) ** 2)
    cnt = orc.score_essential(E, x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1], thr)
    bi, _ = orc.select_best(cnt)
    mask = orc.inlier_mask_essential(E[bi], x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1], thr)
    fr = synthetic.pnp_frames(B=1, n=200, seed=3)
    P = np.stack([synthetic.projection_matrix(Ph[:, :3], Ph[:, 3]) for Ph in
                  synthetic.perturbed_poses(rng, fr["R"][0], fr["t"][0], H)]).astype(np.float32)
    pc = orc.score_pnp(P, *(fr[k][0] for k in "XYZuv"), np.float32(64.0))
    pm = orc.inlier_mask_pnp(P[0], *(fr[k][0] for k in "XYZuv"), np.float32(64.0))
    P1 = synthetic.projection_matrix(np.eye(3), np.zeros(3), pixels=False).astype(np.float32)
    P2 = synthetic.projection_matrix(R, t, pixels=False).astype(np.float32)
    xyz, front = orc.triangulate(P1, P2, x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1])
    np.savez_compressed(os.path.join(HERE, "scoring.npz"), x1=x1, x2=x2, E=E, thr_sq=thr, e_count=cnt, e_mask=mask,
                        X=fr["X"][0], Y=fr["Y"][0], Z=fr["Z"][0], u=fr["u"][0], v=fr["v"][0], P=P, p_count=pc, p_mask=pm,
                        P1=P1, P2=P2, xyz=xyz, front=front, X_true=X.astype(np.float32))


def board():
    views = synthetic.board_views(3, seed=5)
    obj_all = orc.board_object_points()
    out = {}
    for k, v in enumerate(views):
        ids = v["ids"].flatten()
        obj = obj_all[ids].reshape(-1, 3)
        img = np.concatenate([c.reshape(4, 2) for c in v["corners"]])
        rvec = orc.rodrigues_inv(v["R"])
        uv, uvi = orc.project_points(obj, rvec, v["t"], orc.K4, orc.DIST_COEFFS)
        err, st = orc.reprojection_error(obj, img, rvec, v["t"], orc.K4, orc.DIST_COEFFS)
        out.update({f"ids{k}": ids, f"img{k}": img, f"rvec{k}": rvec, f"tvec{k}": v["t"], f"uv{k}": uv, f"uvi{k}": uvi,
                    f"stats{k}": np.array([st["mean"], st["std"], st["max"], st["min"], st["num_points"]])})
    np.savez_compressed(os.path.join(HERE, "board.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # regenerate only the named fixtures, e.g. `make_golden.py rodrigues`
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        quaternions(); rodrigues(); matching(); scoring(); board()
    print("golden fixtures written to", HERE)
