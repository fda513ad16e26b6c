"""Device hypothesis generation: Philox minimal sampling and the 5-point solver.

Bars: sample indices bit-exact against the NumPy restatement (identical sample sets on CPU and
GPU, north_star); solver solutions are float64-computed unit-norm E stored as f32 — every device
solution must satisfy the essential-matrix constraints and have a counterpart in the NumPy
solver's solution set within 1e-5 (Frobenius, up to sign), and on noise-free samples the true E
must be among them.  No reference code exists for either: parity unpinned.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fivepoint as fp
from pose_estimation_amd import ransac, synthetic

pytestmark = pytest.mark.gpu
_SEEDS = list(range(1 + int(os.environ.get("POSEHOT_SOLVER_SOAK", "0"))))     # seed 0 = the committed case; more for a one-off soak


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def test_philox_known_answer():
    # Random123 known-answer vectors for Philox4x32-10
    assert fp.philox4x32_10(np.zeros(4, np.uint32), np.zeros(2, np.uint32)).tolist() == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    ff = np.full(4, 0xFFFFFFFF, np.uint32)
    assert fp.philox4x32_10(ff, ff[:2]).tolist() == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]


def test_sample_minimal_bit_exact(device):
    n_valid = np.array([100, 5, 4, 8192, 7], np.int32)
    S = 300
    got = ransac.sample_minimal(1234, _t(n_valid, device), 8192, S).cpu().numpy()
    for b, nv in enumerate(n_valid):
        ref = fp.sample_minimal(1234, b, int(nv), S)
        assert np.array_equal(got[b], ref)
        if nv >= 5:
            assert (np.sort(got[b], axis=1)[:, 1:] != np.sort(got[b], axis=1)[:, :-1]).all()   # distinct
            assert got[b].min() >= 0 and got[b].max() < nv
    assert (got[2] == -1).all()
    assert len(np.unique(got[3][:, 0])) > 250        # spread over the range


def _scene(seed, n, noise=0.0, yaw=-10.0):
    rng = np.random.default_rng(seed)
    R = synthetic.rot_y(np.deg2rad(yaw)); t = -R @ np.array([0.5, 0.05, 0.1])
    X = rng.uniform(-1, 1, (n, 3)) + [0, 0, 4.0]
    x1 = X[:, :2] / X[:, 2:3] + rng.normal(0, noise, (n, 2))
    Xc = X @ R.T + t
    x2 = Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, noise, (n, 2))
    E = synthetic.essential_from_pose(R, t)
    return x1.astype(np.float32), x2.astype(np.float32), E / np.linalg.norm(E)


@pytest.mark.parametrize("seed", _SEEDS)
@pytest.mark.parametrize("noise,yaw", [(0.0, -10.0), (1.0 / 3108.95, -10.0), (0.0, 120.0)])
def test_fivepoint_matches_numpy_and_constraints(device, noise, yaw, seed):
    n, S = 500, 200
    x1, x2, Et = _scene(3 + 17 * seed, n, noise, yaw)
    nv = _t(np.array([n], np.int32), device)
    samples = ransac.sample_minimal(7 + seed, nv, n, S)
    args = [_t(a[None], device) for a in (x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1])]
    E, ns = ransac.solve_essential_5pt(*args, samples, want_counts=True)
    E = E.cpu().numpy()[0].reshape(S, 10, 3, 3).astype(np.float64); ns = ns.cpu().numpy()[0]
    si = samples.cpu().numpy()[0]
    a = [v.astype(np.float64) for v in (x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1])]
    ref = fp.solve(*(v[si] for v in a))
    found_true, found_true_ref, matched, total, same_count = 0, 0, 0, 0, 0
    over, over_ref, worst = 0, 0, 0.0
    for s in range(S):
        k = int(ns[s])
        sol = E[s][np.abs(E[s]).sum(axis=(1, 2)) > 0]          # slot = root index; non-real roots are zero slots
        assert len(sol) == k
        same_count += (k == len(ref[s]))
        for e in ref[s]:
            _, cub, det = fp.essential_residuals(e, *(v[si[s]] for v in a))
            over_ref += not (cub < 1e-4 and det < 1e-4)
        for e in sol:
            assert abs(np.linalg.norm(e) - 1) < 1e-6
            epi, cub, det = fp.essential_residuals(e, *(v[si[s]] for v in a))
            assert epi < 1e-5                                        # E lies in the sample's null space by construction
            over += not (cub < 1e-4 and det < 1e-4)                  # f32 storage; near-double roots are the loosest
            worst = max(worst, cub, det)
            total += 1
            if len(ref[s]):
                d = np.minimum(np.linalg.norm(ref[s] - e, axis=(1, 2)), np.linalg.norm(ref[s] + e, axis=(1, 2))).min()
                matched += d < 1e-5
        if len(ref[s]) and noise == 0.0:
            found_true_ref += min(np.linalg.norm(ref[s] - Et, axis=(1, 2)).min(), np.linalg.norm(ref[s] + Et, axis=(1, 2)).min()) < 1e-3
        if k and noise == 0.0:
            found_true += min(np.linalg.norm(sol - Et, axis=(1, 2)).min(), np.linalg.norm(sol + Et, axis=(1, 2)).min()) < 1e-3   # inputs are f32-rounded points
    print(f"SOAK5 seed {seed} noise {noise:.1e} yaw {yaw} solutions {total} over_bar {over} over_bar_numpy {over_ref} worst {worst:.1e} "
          f"matched {matched} same_count {same_count}/{S}")
    # The committed case: every solution meets the constraint bars.  Soak seeds (POSEHOT_SOLVER_SOAK): about one minimal sample in
    # 4 000 is ill-conditioned for Nister's elimination (which one depends on the null-space basis, so the NumPy solver's are other
    # samples, at the same rate) and its solutions carry constraint residuals of 1e-4 .. 1e-2: a rate, not a per-solution bar.
    assert over == 0 if seed == 0 else over <= 0.005 * total
    assert total > 2 * S                     # several real solutions per sample on average
    assert matched >= 0.98 * total           # same solution set as the NumPy solver (near-double roots may differ)
    assert same_count >= 0.95 * S
    if noise == 0.0:
        # the true E is among the solutions wherever the float64 NumPy solver finds it on the same f32 points
        assert found_true_ref >= 0.9 * S and found_true >= found_true_ref - 0.01 * S, (found_true, found_true_ref)


def test_fivepoint_degenerate_and_invalid_samples(device):
    x = np.zeros((1, 10), np.float32)
    samples = np.array([[[0, 1, 2, 3, 4], [-1, -1, -1, -1, -1]]], np.int32)
    E, ns = ransac.solve_essential_5pt(*(_t(x, device),) * 4, _t(samples, device), want_counts=True)
    assert torch.isfinite(E).all() and int(ns.sum()) == 0 and float(E.abs().sum()) == 0.0


def test_generated_hypotheses_win_ransac(device):
    """End to end on 30 % outliers: sample -> solve -> score -> best is within 0.1 degrees of the truth."""
    from pose_estimation_amd import geometry
    n = 4000
    x1, x2, Et = _scene(11, n, 1.0 / 3108.95)
    rng = np.random.default_rng(0)
    bad = rng.random(n) < 0.3
    x2[bad] = rng.uniform(-0.3, 0.3, (bad.sum(), 2)).astype(np.float32)
    nv = _t(np.array([n], np.int32), device)
    samples = ransac.sample_minimal(99, nv, n, 410)
    args = [_t(a[None], device) for a in (x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1])]
    E = ransac.solve_essential_5pt(*args, samples)
    thr_sq = np.float32((1.5 / 3108.95) ** 2)
    cnt = ransac.score_essential(E, *args, thr_sq)
    bi, bc = ransac.select_best(cnt)
    assert int(bc[0]) > 0.6 * (~bad).sum()
    Eb = E[0, int(bi[0])].cpu().numpy().reshape(3, 3).astype(np.float64)
    R = synthetic.rot_y(np.deg2rad(-10.0)); t = -R @ np.array([0.5, 0.05, 0.1])
    Rc, tc = geometry.decompose_essential(Eb[None])
    err = min(geometry.rotation_angle_deg(Rc[0, c], R) + geometry.direction_angle_deg(tc[0, c], t) for c in range(4))
    assert err < 0.5      # a minimal-sample model (5 noisy points), before any refinement


# ------------------------------------------------------------------------------------------- P3P
from oracle import p3p as op  # noqa: E402


def _pnp_scene(seed, n, noise=0.0):
    rng = np.random.default_rng(seed)
    R = synthetic.rot_y(np.deg2rad(25.0)) @ synthetic.rot_x(np.deg2rad(-15.0))
    t = np.array([0.2, -0.1, 4.0])
    X = rng.uniform(-1, 1, (n, 3))
    Xc = X @ R.T + t
    xy = Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, noise, (n, 2))
    return X.astype(np.float32), xy.astype(np.float32), R, t


@pytest.mark.parametrize("seed", _SEEDS)
@pytest.mark.parametrize("noise", [0.0, 1.0 / 3108.95])
def test_p3p_solutions_are_solutions_and_match_numpy(device, noise, seed):
    """Every device P3P output is a rotation + translation that maps the three sample points onto their bearings
    with positive depth (the definition of a P3P solution, checked in float64 by the oracle), the solution set
    agrees with the NumPy solver (np.roots + Kabsch) and, on noise-free data, contains the true pose."""
    n, S = 400, 300
    X, xy, R, t = _pnp_scene(11 + 17 * seed, n, noise)
    nv = _t(np.array([n], np.int32), device)
    samples = ransac.sample_minimal(5 + seed, nv, n, S, m=3)
    P, ns = ransac.solve_p3p(*(_t(X[:, k][None], device) for k in range(3)), _t(xy[:, 0][None], device),
                             _t(xy[:, 1][None], device), samples, want_counts=True)
    P, ns, smp = P.cpu().numpy()[0].reshape(S, 4, 3, 4), ns.cpu().numpy()[0], samples.cpu().numpy()[0]
    f_all = op.bearings(xy)
    found = matched = total = same_count = over = n_sol = 0
    worst = 0.0
    for s in range(S):
        Xs, fs = X[smp[s]].astype(np.float64), f_all[smp[s]]
        sols = [p for p in P[s] if np.any(p != 0)]
        assert len(sols) == ns[s] and ns[s] <= 4
        for p in sols:
            orth, det, depth, mism = op.check_solution(p[:, :3].astype(np.float64), p[:, 3].astype(np.float64), Xs, fs)
            assert orth < 1e-5 and abs(det - 1) < 1e-5 and depth > 0
            over += not mism < 2e-5                                                       # f32 storage of an f64 result
            worst = max(worst, mism)
            n_sol += 1
        ref = op.solve_p3p(Xs, fs)
        same_count += len(ref) == len(sols)
        for Rr, tr in ref:
            total += 1
            matched += any(np.abs(p[:, :3] - Rr).max() < 1e-4 and np.abs(p[:, 3] - tr).max() < 1e-4 * max(1, np.abs(tr).max())
                           for p in sols)
        found += any(np.abs(p[:, :3] - R).max() < 2e-3 and np.abs(p[:, 3] - t).max() < 2e-2 for p in sols)
    print(f"SOAK3 seed {seed} noise {noise:.1e} solutions {n_sol} over_bar {over} worst {worst:.1e} matched {matched}/{total} "
          f"same_count {same_count}/{S}")
    # soak seeds: a sample with a point a few millimetres from the camera centre reproduces its bearings less tightly (a rate)
    assert over == 0 if seed == 0 else over <= 0.005 * n_sol
    assert matched >= 0.99 * total and same_count >= 0.97 * S
    if noise == 0.0:
        assert found >= 0.99 * S     # f32 inputs: the true pose is reproduced to input precision


def test_p3p_camera_matrix_output_and_degenerate_samples(device):
    """K4 given: P = K [R|t], so ph_score_pnp on pixels counts every noise-free point as an inlier for the best
    hypothesis.  Samples with an invalid index (-1) or coincident points give all-zero slots and count 0."""
    n, S = 300, 64
    X, xy, R, t = _pnp_scene(12, n)
    K4 = np.array([3108.95, 3101.48, 1129.30, 1940.89])
    pix = (xy.astype(np.float64) * K4[:2] + K4[2:]).astype(np.float32)
    nv = _t(np.array([n], np.int32), device)
    samples = ransac.sample_minimal(9, nv, n, S, m=3).clone()
    samples[0, 0, 1] = -1
    samples[0, 1, 2] = samples[0, 1, 0]
    d = lambda a: _t(a[None], device)
    Xd, Yd, Zd = d(X[:, 0]), d(X[:, 1]), d(X[:, 2])
    P, ns = ransac.solve_p3p(Xd, Yd, Zd, d(xy[:, 0]), d(xy[:, 1]), samples, K4=_t(K4, device), want_counts=True)
    assert ns[0, 0].item() == 0 and ns[0, 1].item() == 0 and not P[0, :8].any().item()
    cnt = ransac.score_pnp(P, Xd, Yd, Zd, d(pix[:, 0]), d(pix[:, 1]), 1.0)
    bi, bc = ransac.select_best(cnt)
    assert bc.item() >= n - 2
    Pb = P[0, bi.item()].cpu().numpy().reshape(3, 4).astype(np.float64)
    Kinv = np.array([[1 / K4[0], 0, -K4[2] / K4[0]], [0, 1 / K4[1], -K4[3] / K4[1]], [0, 0, 1]])
    Rt = Kinv @ Pb
    assert np.abs(Rt[:, :3] - R).max() < 2e-3 and np.abs(Rt[:, 3] - t).max() < 2e-2
    assert cnt[0, :8].sum().item() == 0
