"""The synthetic scenes the benchmarks and GPU tests are built on — checked on the CPU: planted
structure, geometry consistent with the stated poses, determinism of the seeds."""
import numpy as np

from pose_estimation_amd import synthetic


def test_two_view_pair_is_consistent_with_its_pose():
    sc = synthetic.two_view_pair(n=512, seed=3, noise_px=0.0)
    pr = sc["partner"]; ok = pr >= 0
    assert 0.5 < ok.mean() < 0.7
    x1 = np.concatenate([sc["kp1"][ok], np.ones((ok.sum(), 1), np.float32)], axis=1).astype(np.float64)
    x2 = np.concatenate([sc["kp2"][pr[ok]], np.ones((ok.sum(), 1), np.float32)], axis=1).astype(np.float64)
    E = synthetic.essential_from_pose(sc["R"], sc["t"])
    res = np.abs(np.einsum("ni,ij,nj->n", x2, E, x1)) / np.linalg.norm(E)
    assert res.max() < 1e-5                                   # planted matches satisfy the epipolar constraint
    d = np.linalg.norm(sc["desc1"][ok] - sc["desc2"][pr[ok]], axis=1)
    assert np.median(d) < 0.7 and np.allclose(np.linalg.norm(sc["desc1"], axis=1), 1.0, atol=1e-5)
    again = synthetic.two_view_pair(n=512, seed=3, noise_px=0.0)
    assert np.array_equal(again["desc1"], sc["desc1"]) and np.array_equal(again["kp2"], sc["kp2"])


def test_binary_and_float_views_share_landmarks_across_views():
    for make in (synthetic.binary_views, synthetic.float_views):
        sc = make(3, 256, seed=11)
        assert sc["desc"].shape[:2] == (3, 256) and sc["kp"].shape == (3, 256, 2)
        common, i0, i1 = np.intersect1d(sc["ids"][0], sc["ids"][1], return_indices=True)
        assert common.size > 64                                # half of the landmarks are seen by each view
        R, t = synthetic.relative_pose(sc["R"][0], sc["t"][0], sc["R"][1], sc["t"][1])
        E = synthetic.essential_from_pose(R, t)
        x1 = np.concatenate([sc["kp"][0][i0], np.ones((common.size, 1))], axis=1).astype(np.float64)
        x2 = np.concatenate([sc["kp"][1][i1], np.ones((common.size, 1))], axis=1).astype(np.float64)
        res = np.abs(np.einsum("ni,ij,nj->n", x2, E, x1)) / np.linalg.norm(E)
        assert np.median(res) < 3.0 / synthetic.FX             # 1 px of keypoint noise
        a, b = sc["desc"][0][i0], sc["desc"][1][i1]
        if a.dtype == np.uint8:                                # 5 % of 256 bits flipped in each view
            ham = np.unpackbits(a ^ b, axis=1).sum(1)
            assert 10 < np.median(ham) < 40
        else:
            assert np.median(np.linalg.norm(a - b, axis=1)) < 0.9
            assert np.allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-5) and (a >= 0).all()


def test_pnp_frames_project_onto_their_pixels():
    fr = synthetic.pnp_frames(2, 1000, outliers=0.3, noise_px=0.0, seed=9)
    for b in range(2):
        X = np.stack([fr[k][b] for k in "XYZ"], axis=1).astype(np.float64)
        Xc = X @ fr["R"][b].T + fr["t"][b]
        u = synthetic.FX * Xc[:, 0] / Xc[:, 2] + synthetic.CX
        v = synthetic.FY * Xc[:, 1] / Xc[:, 2] + synthetic.CY
        ok = fr["inlier_truth"][b]
        assert 0.6 < ok.mean() < 0.8
        assert np.abs(u[ok] - fr["u"][b][ok]).max() < 1e-2 and np.abs(v[ok] - fr["v"][b][ok]).max() < 1e-2
