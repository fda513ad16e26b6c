"""GPU parity of hypothesis scoring / inlier masks against the CPU oracle.

Bar: inlier counts and masks bit-exact (identical hypotheses, identical points: the f32 op
sequence is the contract of DESIGN.md "Arithmetic contracts").
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from pose_estimation_amd import ransac, synthetic

pytestmark = pytest.mark.gpu


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _essential_case(seed, n, H, outlier_frac=0.4):
    """Correspondences of a known relative pose + E hypotheses (truth first, then perturbed)."""
    rng = np.random.default_rng(seed)
    R = synthetic.rot_y(np.deg2rad(-10.0)); t = -R @ np.array([0.5, 0.0, 0.0])
    X = rng.uniform(-1, 1, (n, 3)) + [0, 0, 4.0]
    x1 = X[:, :2] / X[:, 2:3]
    Xc = X @ R.T + t
    x2 = Xc[:, :2] / Xc[:, 2:3]
    x1 = x1 + rng.normal(0, 1 / synthetic.FX, x1.shape); x2 = x2 + rng.normal(0, 1 / synthetic.FX, x2.shape)
    bad = rng.random(n) < outlier_frac
    x2[bad] = rng.uniform(-0.3, 0.3, (bad.sum(), 2))
    E = np.empty((H, 9), np.float32)
    for h in range(H):
        if h == 0:
            Rh, th = R, t
        else:
            w = rng.normal(0, 0.01, 3)
            Rh = (np.eye(3) + synthetic.skew(w)) @ R
            th = t + rng.normal(0, 0.02, 3)
        E[h] = synthetic.essential_from_pose(Rh, th).reshape(9)
    return x1.astype(np.float32), x2.astype(np.float32), E, bad


@pytest.mark.parametrize("n,H", [(1, 1), (63, 5), (64, 8), (1000, 33), (5000, 257), (8192, 1024)])
def test_score_essential_bit_exact(device, n, H):
    x1, x2, E, _ = _essential_case(100 + n, n, H)
    thr_sq = np.float32((1.5 / synthetic.FX) ** 2)
    cnt = ransac.score_essential(_t(E[None], device), _t(x1[None, :, 0], device), _t(x1[None, :, 1], device),
                                 _t(x2[None, :, 0], device), _t(x2[None, :, 1], device), thr_sq).cpu().numpy()[0]
    ref = orc.score_essential(E, x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1], thr_sq)
    assert np.array_equal(cnt, ref)
    if n >= 1000:
        assert cnt[0] == cnt.max() and cnt[0] > 0.4 * n


def test_score_essential_batched_ragged_and_mask(device):
    B, N, H = 4, 3000, 70
    cases = [_essential_case(7 + b, N, H) for b in range(B)]
    n_valid = np.array([3000, 1, 0, 2047], np.int32)
    x1 = np.stack([c[0] for c in cases]); x2 = np.stack([c[1] for c in cases]); E = np.stack([c[2] for c in cases])
    thr_sq = np.float32((2.0 / synthetic.FX) ** 2)
    args = (_t(x1[:, :, 0], device), _t(x1[:, :, 1], device), _t(x2[:, :, 0], device), _t(x2[:, :, 1], device))
    cnt = ransac.score_essential(_t(E, device), *args, thr_sq, _t(n_valid, device))
    bi, bc = ransac.select_best(cnt)
    mask = ransac.inlier_mask_essential(_t(E, device), bi, *args, thr_sq, _t(n_valid, device))
    cnt, bi, bc, mask = cnt.cpu().numpy(), bi.cpu().numpy(), bc.cpu().numpy(), mask.cpu().numpy().view(np.uint64)
    for b in range(B):
        nv = n_valid[b]
        ref = orc.score_essential(E[b], x1[b, :nv, 0], x1[b, :nv, 1], x2[b, :nv, 0], x2[b, :nv, 1], thr_sq)
        assert np.array_equal(cnt[b], ref)
        rbi, rbc = orc.select_best(ref)
        assert (bi[b], bc[b]) == (rbi, rbc)
        rmask = orc.inlier_mask_essential(E[b, rbi], x1[b, :nv, 0], x1[b, :nv, 1], x2[b, :nv, 0], x2[b, :nv, 1], thr_sq, N)
        assert np.array_equal(mask[b], rmask)
        assert orc.unpack_mask(rmask, N).sum() == rbc


def test_select_best_ties_lowest_index(device):
    c = np.array([[3, 9, 9, 1], [0, 0, 0, 0], [5, 4, 3, 5]], np.int32)
    bi, bc = ransac.select_best(_t(c, device))
    assert bi.cpu().tolist() == [1, 0, 0] and bc.cpu().tolist() == [9, 0, 5]


@pytest.mark.parametrize("n,H", [(1, 1), (100, 9), (4096, 64), (20000, 100)])
def test_score_pnp_bit_exact(device, n, H):
    fr = synthetic.pnp_frames(B=2, n=n, seed=50 + n)
    rng = np.random.default_rng(n)
    P = np.stack([np.stack([synthetic.projection_matrix(Ph[:, :3], Ph[:, 3])
                            for Ph in synthetic.perturbed_poses(rng, fr["R"][b], fr["t"][b], H)]) for b in range(2)]).astype(np.float32)
    thr_sq = np.float32(8.0 ** 2)
    a = [_t(fr[k], device) for k in "XYZuv"]
    cnt = ransac.score_pnp(_t(P, device), *a, thr_sq)
    bi, bc = ransac.select_best(cnt)
    mask = ransac.inlier_mask_pnp(_t(P, device), bi, *a, thr_sq).cpu().numpy().view(np.uint64)
    cnt = cnt.cpu().numpy()
    for b in range(2):
        ref = orc.score_pnp(P[b], *(fr[k][b] for k in "XYZuv"), thr_sq)
        assert np.array_equal(cnt[b], ref)
        rbi, _ = orc.select_best(ref)
        assert int(bi[b]) == rbi
        rmask = orc.inlier_mask_pnp(P[b, rbi], *(fr[k][b] for k in "XYZuv"), thr_sq)
        assert np.array_equal(mask[b], rmask)
    if n >= 4096:  # the true pose (hypothesis 0) explains the non-outlier 70 %
        assert cnt[0, 0] == cnt[0].max() and abs(cnt[0, 0] / n - 0.7) < 0.05


def test_score_pnp_behind_camera_is_outlier(device):
    P = np.array([[[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]]], np.float32)
    X = np.array([[0.0, 0.0]], np.float32); Z = np.array([[2.0, -2.0]], np.float32)
    z = _t(X, device)
    cnt = ransac.score_pnp(_t(P, device), z, z, _t(Z, device), z, z, 1.0).cpu().numpy()
    assert cnt[0, 0] == 1


def test_compact_models_and_lookup(device):
    """ph_compact_models: live models first in slot order, zero tail, slot map; scoring the compact block and mapping the
    winner back gives the winner of the slot layout (ties -> lowest slot), for E (stride 9) and P (stride 12)."""
    from pose_estimation_amd import ransac
    rng = np.random.default_rng(12)
    for stride, H in ((9, 2500), (12, 1030), (9, 7)):
        B = 3
        M = rng.normal(size=(B, H, stride)).astype(np.float32)
        dead = rng.random((B, H)) < 0.55
        dead[1] = True; dead[2, :] = False                   # an item without any live model, one without an empty slot
        M[dead] = 0.0
        M[0, 5] = M[0, 3]                                    # a tie between two live slots
        M[0, 1, 0] = -0.0 if dead[0, 1] else M[0, 1, 0]      # a negative zero is still an empty slot
        Mc, smap, nl = ransac.compact_models(torch.from_numpy(M).to(device))
        Mc, smap, nl = Mc.cpu().numpy(), smap.cpu().numpy(), nl.cpu().numpy()
        for b in range(B):
            live = np.abs(M[b]).sum(axis=1) > 0
            n = int(live.sum())
            assert nl[b] == n and np.array_equal(Mc[b, :n], M[b][live]) and not Mc[b, n:].any()
            assert np.array_equal(smap[b, :n], np.nonzero(live)[0]) and (smap[b, n:] == -1).all()
        sel = torch.tensor([0, 5, H], dtype=torch.int32, device=device)
        got = ransac.lookup_index(torch.from_numpy(smap).to(device), sel).cpu().numpy()
        assert got[0] == smap[0, 0] and got[1] == -1 and got[2] == -1
        # arg-max and lookup in one launch (ph_select_best_mapped) = the two calls
        cnt = torch.from_numpy(rng.integers(0, 50, (B, H)).astype(np.int32)).to(device)
        bi_c, bc = ransac.select_best(cnt)
        want = ransac.lookup_index(torch.from_numpy(smap).to(device), bi_c)
        bi_m, bc_m = ransac.select_best(cnt, torch.from_numpy(smap).to(device))
        assert torch.equal(bi_m, want) and torch.equal(bc_m, bc)


@pytest.mark.parametrize("stride,group,S", [(9, 10, 920), (9, 10, 1), (9, 10, 129), (12, 4, 4096), (12, 4, 321), (9, 5, 100)])
def test_compact_models_counted_equals_the_plain_compaction(device, stride, group, S):
    """ph_compact_models_counted (the solvers' per-sample counts place the pieces of an item: many blocks per item) gives
    what ph_compact_models gives — compact block, slot map, n_live — for both solver shapes, sizes that end inside a chunk,
    items without a live model / without an empty slot; a (stride, group) without a kernel of its own takes the plain path."""
    from pose_estimation_amd import ransac
    rng = np.random.default_rng(100 * stride + S)
    B, H = 4, S * group
    M = rng.normal(size=(B, H, stride)).astype(np.float32)
    dead = rng.random((B, H)) < 0.55
    dead[1] = True; dead[2] = False
    M[dead] = 0.0
    counts = (~dead).reshape(B, S, group).sum(axis=2).astype(np.int32)
    d = torch.from_numpy(M).to(device)
    ref = [a.cpu().numpy() for a in ransac.compact_models(d)]
    got = [a.cpu().numpy() for a in ransac.compact_models(d, counts=torch.from_numpy(counts).to(device))]
    for r, g in zip(ref, got):
        assert np.array_equal(r.view(np.uint32) if r.dtype == np.float32 else r, g.view(np.uint32) if g.dtype == np.float32 else g)
    assert np.array_equal(got[2], (~dead).sum(axis=1))
