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


STDOUT_SEED = 21         # seed of the synthetic ring behind the stdout goldens (chosen by `best_stdout_seed`)


def stdout_texts(seed=None, write=True):
    """Expected stdout of `main.py --detections D` and `validate_pose.py --detections D` for the synthetic 8-camera
    detections file (SURVEY.md §8d cfg 1 substitute: `synthetic.board_views(8)`), written line by line from the
    reference's own print formats — /root/reference/main.py:159,165,184,198-199,441-463,468-484,509,431,512 and
    /root/reference/validate_pose.py:303-305,314,333,339,357,374-375,391-407 — with the numbers from the CPU path
    (host homography initial pose -> the oracle's NumPy Levenberg-Marquardt -> the oracle's SciPy-pinned camera
    inversion / reprojection statistics), NOT from the GPU library: the GPU test compares the entry points' whole
    stdout against these files byte for byte.  `{HTML}` stands for the scene file's path.
    Lines the reference prints and this build does not: the per-image overlay files ("  💾 Saved: ...",
    validate_pose.py:387) and the closing two lines about them (validate_pose.py:409-410) — drawing is out of scope."""
    from pose_estimation_amd import board as bd
    views = synthetic.board_views(8, seed=STDOUT_SEED if seed is None else seed)
    _, board = bd.create_aruco_board()
    W = (bd.MARKERS_X * bd.MARKER_LENGTH_MM + (bd.MARKERS_X - 1) * bd.MARKER_SEPARATION_MM) / 1000.0
    H = (bd.MARKERS_Y * bd.MARKER_LENGTH_MM + (bd.MARKERS_Y - 1) * bd.MARKER_SEPARATION_MM) / 1000.0
    worst = [1.0]

    def margin(x, digits):       # distance of x from a rounding boundary at `digits` decimals, in units of the last digit
        f = abs(x) * 10 ** digits
        worst[0] = min(worst[0], abs((f % 1.0) - 0.5))

    rec = []
    for v in views:
        obj, img = board.matchImagePoints(v["corners"], v["ids"])
        obj, img = obj.reshape(-1, 3).astype(np.float64), img.reshape(-1, 2).astype(np.float64)
        p0 = bd.initial_board_pose(obj, img, bd.CAMERA_MATRIX, bd.DIST_COEFFS)
        pose, _ = orc.refine_pose_lm_np(obj, img, p0, max_iters=100)
        pos, fwd, up, Rcw, q = orc.camera_position_and_orientation(pose[:3], pose[3:])
        err, st = orc.reprojection_error(obj.astype(np.float32), img.astype(np.float32), pose[:3], pose[3:], orc.K4, orc.DIST_COEFFS)
        rec.append(dict(filename=v["filename"], ids=v["ids"], pos=pos, q=q, n=len(v["ids"]), st=st))
        for x in pos: margin(x, 3)
        for x in q: margin(x, 4)
        for k in ("mean", "max", "std"): margin(st[k], 2)
    m = ["=" * 80, "🎯 ArUco Board Camera Pose Estimation", "=" * 80, "", "📋 Board Configuration:",
         f"  - Grid: {bd.MARKERS_X} × {bd.MARKERS_Y} markers", f"  - Marker size: {bd.MARKER_LENGTH_MM} mm",
         f"  - Separation: {bd.MARKER_SEPARATION_MM} mm", f"  - Total size: {W*1000:.1f} × {H*1000:.1f} mm",
         "  - Dictionary: DICT_6X6_250", f"📷 Found {len(rec)} images"]
    for r in rec:
        m += ["", f"🔍 Processing: {r['filename']}", f"  ✓ Detected {r['n']} markers: {r['ids'].flatten().tolist()}",
              f"  ✓ Camera position: [{r['pos'][0]:.3f}, {r['pos'][1]:.3f}, {r['pos'][2]:.3f}] m",
              f"  ✓ Quaternion (w,x,y,z): [{r['q'][0]:.4f}, {r['q'][1]:.4f}, {r['q'][2]:.4f}, {r['q'][3]:.4f}]"]
    m += ["", "=" * 100, "📊 CAMERA POSE SUMMARY", "=" * 100,
          f"{'Image':<25} {'#':>4} {'X(m)':>8} {'Y(m)':>8} {'Z(m)':>8} │ {'qw':>7} {'qx':>7} {'qy':>7} {'qz':>7}", "-" * 100]
    for r in rec:
        pos, q = r["pos"], r["q"]
        m.append(f"{r['filename']:<25} {r['n']:>4} {pos[0]:>8.3f} {pos[1]:>8.3f} {pos[2]:>8.3f} │ {q[0]:>7.4f} {q[1]:>7.4f} {q[2]:>7.4f} {q[3]:>7.4f}")
    m += ["-" * 100, "", "📐 RELATIVE DISTANCES BETWEEN CAMERAS (meters):", "-" * 80]
    for i in range(len(rec)):
        for j in range(i + 1, len(rec)):
            d = np.linalg.norm(rec[i]["pos"] - rec[j]["pos"])
            margin(d, 3)
            m.append(f"  Cam {i+1} ↔ Cam {j+1}: {d:.3f} m")
    m += ["", "🎨 Creating interactive 3D visualization...", "", "✅ Interactive 3D visualization saved to: {HTML}", "", "✅ Done!"]
    if write:
        open(os.path.join(HERE, "main_stdout.txt"), "w", encoding="utf-8").write("\n".join(m) + "\n")

    v_ = ["=" * 80, "🔍 Pose Estimation Validation", "=" * 80, "", f"📋 Board: {W*1000:.1f} × {H*1000:.1f} mm",
          f"📷 Found {len(rec)} images", ""]
    means = []
    for r in rec:
        st = r["st"]
        v_ += [f"🔍 Validating: {r['filename']}", f"  ✓ Detected {r['n']} markers",
               f"  📊 Reprojection Error: mean={st['mean']:.2f}px, max={st['max']:.2f}px, std={st['std']:.2f}px"]
        means.append(st["mean"])
    for x in (np.mean(means), np.max(means), np.min(means), np.std(means)): margin(x, 2)
    grade = orc.summary_of_means(means)["grade"]
    assert grade == "EXCELLENT"
    v_ += ["", "=" * 80, "📊 VALIDATION SUMMARY", "=" * 80, f"  Mean reprojection error: {np.mean(means):.2f} pixels",
           f"  Max reprojection error:  {np.max(means):.2f} pixels", f"  Min reprojection error:  {np.min(means):.2f} pixels",
           f"  Std deviation:           {np.std(means):.2f} pixels", "-" * 80,
           "  ✅ EXCELLENT: Mean error < 1 pixel - Pose estimation is very accurate!"]
    if write:
        open(os.path.join(HERE, "validate_stdout.txt"), "w", encoding="utf-8").write("\n".join(v_) + "\n")
    else:
        return worst[0]
    # every printed number sits at least this far (in units of its last printed digit) from a rounding boundary: the GPU
    # path agrees with the CPU path to ~1e-6 of a unit, so the texts are stable
    print(f"stdout goldens: smallest distance of a printed number from a rounding boundary = {worst[0]:.4f} last-digit units")
    assert worst[0] > 5e-3, "a printed value sits on a rounding boundary: run best_stdout_seed and change STDOUT_SEED"


def best_stdout_seed():
    """Seed (of the first 40) whose printed numbers stay farthest from a rounding boundary."""
    best = max(range(1, 41), key=lambda sd: stdout_texts(seed=sd, write=False))
    print("best seed", best, "margin", stdout_texts(seed=best, write=False))


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # regenerate only the named fixtures, e.g. `make_golden.py rodrigues`
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        quaternions(); rodrigues(); matching(); scoring(); board(); stdout_texts()
    print("golden fixtures written to", HERE)
