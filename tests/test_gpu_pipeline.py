"""Whole relative-pose pipeline on the GPU against the CPU oracle, with identical RANSAC sample sets.

north_star bars: identical sample sets (counter-based Philox) -> inlier masks bit-exact; rotation /
translation angular difference between the CPU path and the GPU path < 0.1 degrees.
CPU path = oracle matcher -> ratio -> compaction -> NumPy Philox samples -> NumPy 5-point solver ->
oracle Sampson scoring.  GPU path = pose_estimation_amd.pipeline.relative_pose_batch.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fivepoint as fp
from oracle import oracle as orc
from pose_estimation_amd import geometry, matching, pipeline, ransac, synthetic

pytestmark = pytest.mark.gpu
_SEEDS = list(range(1 + int(os.environ.get("POSEHOT_PIPELINE_SOAK", "0"))))    # seed 0 = the committed case; more for a one-off soak


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _best_pose_err(E9, R, t):
    Rc, tc = geometry.decompose_essential(np.asarray(E9, np.float64).reshape(1, 3, 3))
    return min(geometry.rotation_angle_deg(Rc[0, c], R) + geometry.direction_angle_deg(tc[0, c], t) for c in range(4))


@pytest.mark.parametrize("soak", _SEEDS)
@pytest.mark.parametrize("decompose", ["device", "host"])
def test_pipeline_matches_cpu_path_with_identical_samples(device, decompose, soak):
    n, S, seed = 2048, 120, 4242 + soak
    scs = [synthetic.two_view_pair(n=n, seed=100 + b + 1000 * soak) for b in range(2)]
    dq = _t(np.stack([s["desc1"] for s in scs]), device); dt = _t(np.stack([s["desc2"] for s in scs]), device)
    kq = _t(np.stack([s["kp1"] for s in scs]), device); kt = _t(np.stack([s["kp2"] for s in scs]), device)
    thr_sq = np.float32((1.0 / synthetic.FX) ** 2)
    out = pipeline.relative_pose_batch(lambda: (*matching.match_l2(dq, dt), "l2"), kq, kt, None, None, None,
                                       float(thr_sq), n_samples=S, seed=seed, want_points=True, decompose=decompose)
    table = out["table"].cpu().numpy()
    E_gpu = out["E_hyp"].cpu().numpy()
    mask_gpu = out["mask"].cpu().numpy().view(np.uint64)
    for b, sc in enumerate(scs):
        # ---- CPU path up to the correspondences: bit-exact with the GPU's
        idx, d1, d2 = orc.match_l2_top2(sc["desc1"], sc["desc2"])
        keep = orc.ratio_filter_l2(idx, d1, d2, 0.75)
        m = orc.compact_matches(idx, keep, sc["kp1"], sc["kp2"])
        nm = m["n_match"]
        assert nm == int(out["n_match"][b]) == int(table[b, 13]) and nm > 500
        pts = [m[k][:nm] for k in ("x1", "y1", "x2", "y2")]
        for k, g in zip(("x1", "y1", "x2", "y2"), out["pts"]):
            assert np.array_equal(g[b, :nm].cpu().numpy(), m[k][:nm])
        # ---- identical sample sets
        samples = fp.sample_minimal(seed, b, nm, S)
        got = ransac.sample_minimal(seed, out["n_match"], n, S).cpu().numpy()[b]
        assert np.array_equal(got, samples)
        # ---- the GPU's hypotheses scored on the CPU: counts, winner and mask bit-exact
        cnt = orc.score_essential(E_gpu[b], *pts, thr_sq)
        bi, bc = orc.select_best(cnt)
        assert (bi, bc) == (int(out["best_idx"][b]), int(out["best_count"][b])) == (int(table[b, 14]), int(table[b, 12]))
        assert np.array_equal(orc.inlier_mask_essential(E_gpu[b, bi], *pts, thr_sq, n), mask_gpu[b])
        # ---- CPU solver on the same samples: same solutions (angular difference far below 0.1 degrees)
        p64 = [p.astype(np.float64) for p in pts]
        ref = fp.solve(*(p[samples] for p in p64))
        cpu_best, cpu_best_cnt, cpu_best_s = None, -1, -1
        for s in range(S):
            if len(ref[s]) == 0:
                continue
            c = orc.score_essential(ref[s].reshape(-1, 9).astype(np.float32), *pts, thr_sq)
            if c.max() > cpu_best_cnt:
                cpu_best_cnt, cpu_best, cpu_best_s = int(c.max()), ref[s][int(c.argmax())], s
        slots = E_gpu[b, 10 * cpu_best_s: 10 * cpu_best_s + 10].reshape(10, 3, 3).astype(np.float64)
        d = np.minimum(np.linalg.norm(slots - cpu_best, axis=(1, 2)), np.linalg.norm(slots + cpu_best, axis=(1, 2)))
        twin = slots[int(d.argmin())]
        Rg, tg = geometry.decompose_essential(twin[None]); Rc, tc = geometry.decompose_essential(cpu_best[None])
        for c in range(4):      # the four (R, t) candidates agree as sets (their order depends on the SVD's signs)
            assert min(max(geometry.rotation_angle_deg(Rg[0, c], Rc[0, k]), geometry.direction_angle_deg(tg[0, c], tc[0, k]))
                       for k in range(4)) < 0.1
        assert abs(cpu_best_cnt - bc) <= max(2, 0.01 * bc)       # both paths find an equally good model
        # ---- and the pose the GPU path reports is the truth to minimal-sample accuracy, cheirality resolved
        assert geometry.rotation_angle_deg(table[b, :9].reshape(3, 3), sc["R"]) < 1.0
        assert geometry.direction_angle_deg(table[b, 9:12], sc["t"]) < 2.0
        assert table[b, 15] == pipeline.STATUS_OK
        # triangulated inliers sit in front of both cameras, ~4 m away
        sel = orc.unpack_mask(mask_gpu[b], n)[:nm]
        Z = out["xyz"][b, :nm].cpu().numpy()[sel, 2]
        scale = np.linalg.norm(sc["t"])                                   # |t| is unobservable: recovered t is unit
        assert np.median(Z) * scale == pytest.approx(4.0, rel=0.25)
        print(f"SOAKP seed {soak} {decompose} pair {b}: matches {nm} best count {bc} (CPU solver's best {cpu_best_cnt}) "
              f"rot err {geometry.rotation_angle_deg(table[b, :9].reshape(3, 3), sc['R']):.3f} deg "
              f"t dir err {geometry.direction_angle_deg(table[b, 9:12], sc['t']):.3f} deg")


def test_pipeline_with_hamming_matcher_and_given_hypotheses(device):
    sc = synthetic.binary_views(2, 4096, seed=8)
    d = _t(sc["desc"], device); kp = _t(sc["kp"], device)
    vq = _t(np.array([0], np.int32), device); vt = _t(np.array([1], np.int32), device)
    R, t = synthetic.relative_pose(sc["R"][0], sc["t"][0], sc["R"][1], sc["t"][1])
    out = pipeline.relative_pose_batch(lambda: (*matching.match_hamming(d, d, vq, vt), "hamming"), kp, kp, vq, vt, None,
                                       float((1.5 / synthetic.FX) ** 2), ratio=0.8, n_samples=200, seed=5)
    tab = out["table"].cpu().numpy()[0]
    assert tab[13] > 1000 and tab[12] > 0.8 * tab[13]
    assert geometry.rotation_angle_deg(tab[:9].reshape(3, 3), R) < 0.5
    assert geometry.direction_angle_deg(tab[9:12], t) < 1.0


def test_device_decomposition_equals_svd_candidates(device):
    """ph_decompose_essential: same four (R, t) candidates as the host's SVD construction (as a set), R orthonormal
    to f32 rounding of E, selection by sel, the all-zero matrix degenerates to (I, 0)."""
    rng = np.random.default_rng(77)
    B, H = 300, 3
    E = np.zeros((B, H, 9), np.float32)
    truth = []
    for b in range(B):
        R = synthetic.rot_x(rng.uniform(-3, 3)) @ synthetic.rot_y(rng.uniform(-3, 3)) @ synthetic.rot_x(rng.uniform(-3, 3))
        t = rng.normal(size=3)
        e = synthetic.essential_from_pose(R, t)
        e = e / np.linalg.norm(e) * (1.0 if b % 2 else -1.0)
        E[b] = rng.normal(size=(H, 9))
        E[b, b % H] = e.reshape(9).astype(np.float32)
        truth.append(E[b, b % H].astype(np.float64).reshape(3, 3))
    E[B - 1] = 0.0
    sel = (np.arange(B) % H).astype(np.int32)
    P1, P2 = ransac.decompose_essential(_t(E, device), _t(sel, device))
    P1, P2 = P1.cpu().numpy().reshape(B, 4, 3, 4), P2.cpu().numpy().reshape(B, 4, 3, 4)
    eye = np.concatenate([np.eye(3), np.zeros((3, 1))], axis=1)
    assert np.array_equal(P1, np.broadcast_to(eye, P1.shape).astype(np.float32))
    for b in range(B - 1):
        Rs, ts = geometry.decompose_essential(truth[b][None])
        for c in range(4):
            R, t = P2[b, c, :, :3].astype(np.float64), P2[b, c, :, 3].astype(np.float64)
            assert np.abs(R @ R.T - np.eye(3)).max() < 1e-6 and abs(np.linalg.det(R) - 1.0) < 1e-6
            assert min(max(np.abs(R - Rs[0, k]).max(), np.abs(t - ts[0, k]).max()) for k in range(4)) < 1e-6
        assert np.array_equal(P2[b, 0, :, :3], P2[b, 1, :, :3]) and np.array_equal(P2[b, 0, :, 3], -P2[b, 1, :, 3])
        assert np.array_equal(P2[b, 2, :, :3], P2[b, 3, :, :3]) and not np.allclose(P2[b, 0, :, :3], P2[b, 2, :, :3])
    assert np.array_equal(P2[B - 1], np.broadcast_to(eye, (4, 3, 4)).astype(np.float32))


def test_pair_without_a_live_hypothesis_is_flagged_not_faulted(device):
    """A pair with fewer matches than a minimal sample (here: 3 valid queries) has no live model: every solver slot is
    empty, the compact winner maps to slot -1 (posehot.h, ph_compact_models), every consumer clamps it, the row's status
    is 0 — and the healthy pair of the same batch is the row it is on its own."""
    n, S, seed = 2048, 64, 99
    scs = [synthetic.two_view_pair(n=n, seed=300 + b) for b in range(2)]
    dq = _t(np.stack([s["desc1"] for s in scs]), device); dt = _t(np.stack([s["desc2"] for s in scs]), device)
    kq = _t(np.stack([s["kp1"] for s in scs]), device); kt = _t(np.stack([s["kp2"] for s in scs]), device)
    thr_sq = float(np.float32((1.0 / synthetic.FX) ** 2))
    key = _t(np.array([7, 8], np.int32), device)
    for fused in (False, True):
        n_q = _t(np.array([n, 3], np.int32), device)
        if fused:
            match = lambda: (*matching.match_l2_ratio(dq, dt, 0.75, n_q=n_q), "l2_keep")
        else:
            match = lambda: (*matching.match_l2(dq, dt, n_q=n_q), "l2")
        out = pipeline.relative_pose_batch(match, kq, kt, None, None, None, thr_sq, n_samples=S, seed=seed, item_key=key,
                                           want_points=True, want_counts=True)
        torch.cuda.synchronize()
        tab = out["table"].cpu().numpy()
        assert int(out["n_match"][1]) <= 3 and int(out["n_solutions"][1].sum()) == 0
        assert tab[1, 15] == pipeline.STATUS_DEGENERATE and tab[1, 12] == 0
        assert int(out["best_idx"][1]) in (-1, 0) and np.isfinite(tab[1, :12]).all()
        # the healthy pair does not notice its neighbour
        alone = pipeline.relative_pose_batch(
            (lambda: (*matching.match_l2_ratio(dq[:1], dt[:1], 0.75), "l2_keep")) if fused else
            (lambda: (*matching.match_l2(dq[:1], dt[:1]), "l2")),
            kq[:1], kt[:1], None, None, None, thr_sq, n_samples=S, seed=seed, item_key=key[:1])
        assert tab[0, 15] == pipeline.STATUS_OK
        assert np.array_equal(tab[0].view(np.uint32), alone["table"].cpu().numpy()[0].view(np.uint32))
