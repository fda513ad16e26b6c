"""The CPU oracle against independent NumPy statements and the committed golden fixtures.

The reference pins nothing for matching / scoring / triangulation ("parity unpinned", SURVEY.md
§8c); what CAN be pinned is checked here: the quaternion convention against SciPy-generated
vectors (the reference's own dependency, main.py:138-141), the board geometry against the
reference's restatement (validate_pose.py:162-176), and the oracle against brute-force NumPy.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from pose_estimation_amd import synthetic

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hamming_oracle_vs_numpy_bruteforce():
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, (60, 32), dtype=np.uint8); t = rng.integers(0, 256, (90, 32), dtype=np.uint8)
    t[80] = t[10]; q[5] = t[10]
    i, d1, d2 = orc.match_hamming_top2(q, t)
    D = np.unpackbits(q[:, None, :] ^ t[None], axis=2).sum(2)
    assert np.array_equal(i, D.argmin(1)) and np.array_equal(d1, D.min(1)) and np.array_equal(d2, np.sort(D, 1)[:, 1])
    assert i[5] == 10 and d1[5] == 0 and d2[5] == 0
    i, d1, d2 = orc.match_hamming_top2(q, t[:1])
    assert (i == 0).all() and (d2 == orc.HAMMING_NONE).all()
    i, d1, d2 = orc.match_hamming_top2(q, t[:0])
    assert (i == -1).all() and (d1 == orc.HAMMING_NONE).all()


def test_l2_oracle_f32_contract_vs_float64():
    rng = np.random.default_rng(1)
    q = synthetic.sift_like(rng, 200, 128); t = synthetic.sift_like(rng, 300, 128)
    i, a, b = orc.match_l2_top2(q, t)
    j, c, d = orc.match_l2_top2_f64(q, t)
    D = ((q[:, None, :].astype(np.float64) - t[None].astype(np.float64)) ** 2).sum(2)
    assert np.array_equal(j, D.argmin(1)) and np.allclose(c, D.min(1), rtol=1e-12)
    assert np.array_equal(i, j)
    assert np.abs(a - c).max() < 2e-6 and np.abs(b - d).max() < 2e-6
    # the f32 chain in numpy, element by element, for a few pairs
    order = np.array([0, 4, 1, 5, 2, 6, 3, 7])
    tn = orc.row_sqnorm(t); qn = orc.row_sqnorm(q)
    for qi in (0, 17, 199):
        acc = np.float32(-0.5) * tn[i[qi]]
        for cidx in range(0, 128, 8):
            for e in order:
                acc = np.float32(np.float64(t[i[qi], cidx + e]) * np.float64(q[qi, cidx + e]) + np.float64(acc))  # fma: one rounding
        assert np.float32(max(0.0, np.float32(np.float64(-2.0) * np.float64(acc) + np.float64(qn[qi])))) == a[qi]


def test_ratio_and_compaction_oracle():
    idx = np.array([3, -1, 0, 2], np.int32)
    d1 = np.array([10, 0, 30, 5], np.int32); d2 = np.array([20, 0, 40, orc.HAMMING_NONE], np.int32)
    assert orc.ratio_filter_hamming(idx, d1, d2, 0.75).tolist() == [1, 0, 0, 0]   # 30 < 0.75*40 is False (strict)
    f1 = np.array([1.0, 1.0, 0.5625], np.float32); f2 = np.array([4.0, np.inf, 1.0], np.float32)
    assert orc.ratio_filter_l2(np.array([0, 1, 2], np.int32), f1, f2, 0.75).tolist() == [1, 0, 0]
    kq = np.arange(8, dtype=np.float32).reshape(4, 2); kt = 10 + np.arange(8, dtype=np.float32).reshape(4, 2)
    m = orc.compact_matches(np.array([3, 0, 1, 2], np.int32), np.array([1, 0, 0, 1], np.int32), kq, kt)
    assert m["n_match"] == 2 and m["match_q"][:2].tolist() == [0, 3] and m["match_t"][:2].tolist() == [3, 2]
    assert m["x2"][:2].tolist() == [16.0, 14.0]


def test_sampson_oracle_vs_float64_formula():
    g = np.load(os.path.join(G, "scoring.npz"))
    x1, x2, E, thr = g["x1"].astype(np.float64), g["x2"].astype(np.float64), g["E"].astype(np.float64), float(g["thr_sq"])
    h1 = np.concatenate([x1, np.ones((len(x1), 1))], 1); h2 = np.concatenate([x2, np.ones((len(x2), 1))], 1)
    for h in range(E.shape[0]):
        Em = E[h].reshape(3, 3)
        Ex1 = h1 @ Em.T; Etx2 = h2 @ Em
        num = (h2 * Ex1).sum(1) ** 2
        den = Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2
        sampson = num / den
        # f32 vs f64 may differ only for residuals within rounding of the threshold
        assert abs(int((sampson < thr).sum()) - int(g["e_count"][h])) <= 1
    assert g["e_count"][0] == g["e_count"].max()


def test_golden_scoring_and_matching_frozen():
    g = np.load(os.path.join(G, "scoring.npz"))
    assert np.array_equal(orc.score_essential(g["E"], g["x1"][:, 0], g["x1"][:, 1], g["x2"][:, 0], g["x2"][:, 1], g["thr_sq"]), g["e_count"])
    bi, _ = orc.select_best(g["e_count"])
    assert np.array_equal(orc.inlier_mask_essential(g["E"][bi], g["x1"][:, 0], g["x1"][:, 1], g["x2"][:, 0], g["x2"][:, 1], g["thr_sq"]), g["e_mask"])
    assert np.array_equal(orc.score_pnp(g["P"], g["X"], g["Y"], g["Z"], g["u"], g["v"], np.float32(64.0)), g["p_count"])
    assert np.array_equal(orc.inlier_mask_pnp(g["P"][0], g["X"], g["Y"], g["Z"], g["u"], g["v"], np.float32(64.0)), g["p_mask"])
    xyz, front = orc.triangulate(g["P1"], g["P2"], g["x1"][:, 0], g["x1"][:, 1], g["x2"][:, 0], g["x2"][:, 1])
    assert np.allclose(xyz, g["xyz"], rtol=1e-6) and front == int(g["front"])
    ok = np.ones(len(xyz), bool); ok[::5] = False          # every 5th correspondence is a planted outlier
    assert np.abs(xyz[ok] - g["X_true"][ok]).max() < 0.05   # 1 px noise at 4 m depth, 0.5 m baseline
    m = np.load(os.path.join(G, "matching.npz"))
    i, d1, d2 = orc.match_hamming_top2(m["hq"], m["ht"])
    assert np.array_equal(i, m["h_idx"]) and np.array_equal(d1, m["h_d1"]) and np.array_equal(d2, m["h_d2"])
    assert i[0] == 4 and d2[0] == 0
    i, a, b = orc.match_l2_top2(m["fq"], m["ft"])
    assert np.array_equal(i, m["l_idx"]) and np.array_equal(a.view(np.uint32), m["l_d1sq"].view(np.uint32))
    assert np.array_equal(b.view(np.uint32), m["l_d2sq"].view(np.uint32)) and i[0] == 3


def test_pnp_oracle_vs_float64_projection():
    g = np.load(os.path.join(G, "scoring.npz"))
    P = g["P"][0].astype(np.float64).reshape(3, 4)
    Xh = np.stack([g["X"], g["Y"], g["Z"], np.ones_like(g["X"])], 1).astype(np.float64)
    p = Xh @ P.T
    e2 = (p[:, 0] / p[:, 2] - g["u"]) ** 2 + (p[:, 1] / p[:, 2] - g["v"]) ** 2
    assert abs(int(((e2 < 64.0) & (p[:, 2] > 0)).sum()) - int(g["p_count"][0])) <= 1


def test_quaternion_convention_pinned_by_scipy():
    """main.py:138-141: scipy as_quat() [x,y,z,w] reordered to [w,x,y,z], sign not canonicalised."""
    d = json.load(open(os.path.join(G, "quaternions.json")))
    assert len(d["cases"]) >= 40
    for c in d["cases"]:
        R = np.array(c["R"]); q = np.array(c["quat_xyzw"])
        got = orc.quat_wxyz_from_matrix(R)
        assert np.allclose(got, [q[3], q[0], q[1], q[2]], atol=1e-12), (got, q)


def test_board_geometry_follows_reference():
    """validate_pose.py:162-176 and main.py:476-477."""
    pts = orc.board_object_points()
    assert pts.shape == (77, 4, 3) and pts.dtype == np.float32
    assert np.allclose(pts[0], [[0, 0, 0], [0.025, 0, 0], [0.025, 0.025, 0], [0, 0.025, 0]])
    assert np.allclose(pts[12, 0], [0.035, 0.035, 0])                 # id = j*11 + i
    assert np.allclose(pts[76, 2], [0.375, 0.235, 0], atol=1e-7)      # far corner = board size
    assert np.array_equal(pts, synthetic.board_object_points())


def test_rodrigues_roundtrip_and_projection_golden():
    rng = np.random.default_rng(3)
    for _ in range(50):
        r = rng.normal(0, 1.2, 3)
        R = orc.rodrigues(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(R), 1)
        assert np.allclose(orc.rodrigues(orc.rodrigues_inv(R)), R, atol=1e-9)
        if np.linalg.norm(r) < np.pi:
            assert np.allclose(orc.rodrigues_inv(R), r, atol=1e-9)
    g = np.load(os.path.join(G, "board.npz"))
    obj_all = orc.board_object_points()
    for k in range(3):
        obj = obj_all[g[f"ids{k}"]].reshape(-1, 3)
        uv, uvi = orc.project_points(obj, g[f"rvec{k}"], g[f"tvec{k}"], orc.K4, orc.DIST_COEFFS)
        assert np.allclose(uv, g[f"uv{k}"], atol=1e-9) and np.array_equal(uvi, g[f"uvi{k}"])
        assert np.array_equal(uvi, uv.astype(np.int32))               # truncation, validate_pose.py:184
        assert np.allclose(uv, orc.project_points_np(obj, g[f"rvec{k}"], g[f"tvec{k}"]), atol=1e-9)
        err, st = orc.reprojection_error(obj, g[f"img{k}"], g[f"rvec{k}"], g[f"tvec{k}"], orc.K4, orc.DIST_COEFFS)
        assert np.allclose([st["mean"], st["std"], st["max"], st["min"], st["num_points"]], g[f"stats{k}"], atol=1e-9)
        assert np.isclose(st["std"], np.std(err)) and np.isclose(st["mean"], np.mean(err))   # ddof = 0
        assert st["mean"] < 1.0


def test_summary_grades():
    assert orc.summary_of_means([0.3, 0.5])["grade"] == "EXCELLENT"
    assert orc.summary_of_means([1.0, 1.5])["grade"] == "GOOD"
    assert orc.summary_of_means([2.0, 4.0])["grade"] == "FAIR"
    assert orc.summary_of_means([5.0])["grade"] == "POOR"


def test_p3p_oracle_recovers_pose_and_solutions_check():
    """The NumPy P3P checker (np.roots + Kabsch): on noise-free samples every solution reprojects the three points
    onto their bearings with positive depth, and the true pose is among them."""
    from oracle import p3p as op
    rng = np.random.default_rng(7)
    R = synthetic.rot_y(np.deg2rad(20.0)) @ synthetic.rot_x(np.deg2rad(-10.0)); t = np.array([0.1, -0.2, 4.0])
    found = 0
    for _ in range(100):
        X = rng.uniform(-1, 1, (3, 3))
        Y = X @ R.T + t
        f = Y / np.linalg.norm(Y, axis=1, keepdims=True)
        sols = op.solve_p3p(X, f)
        assert 1 <= len(sols) <= 4
        for Rs, ts in sols:
            orth, det, depth, mism = op.check_solution(Rs, ts, X, f)
            assert orth < 1e-9 and abs(det - 1) < 1e-9 and depth > 0 and mism < 1e-7
        found += any(np.abs(Rs - R).max() < 1e-6 and np.abs(ts - t).max() < 1e-6 for Rs, ts in sols)
    assert found >= 97


def test_lm_oracle_converges_to_truth_on_board_views():
    """The NumPy LM checker of ph_refine_pose_lm: from a perturbed start it returns to the true board pose
    (noise-free corners -> rms ~ 0) and lands in the EXCELLENT band with 0.3 px corner noise."""
    for noise, tol_rms in ((0.0, 1e-4), (0.3, 1.0)):   # f32 corner coordinates bound the noise-free residual
        v = synthetic.board_views(1, seed=5, noise_px=noise)[0]
        obj = orc.board_object_points()[v["ids"].flatten()].reshape(-1, 3)
        img = np.concatenate([c.reshape(4, 2) for c in v["corners"]])
        p_true = np.concatenate([orc.rodrigues_inv(v["R"]), v["t"]])
        p0 = p_true + np.array([0.04, -0.03, 0.05, 0.02, -0.01, 0.03])
        p, rms = orc.refine_pose_lm_np(obj, img, p0, max_iters=100)
        assert rms < tol_rms
        assert np.abs(p - p_true).max() < (1e-6 if noise == 0.0 else 2e-3)
