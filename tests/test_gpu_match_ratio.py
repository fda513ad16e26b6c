"""ph_match_l2_ratio (matching + ratio test in one call, exact chains skipped where the approximate scores decide) against
the two-call path: ph_match_l2_top2 (bit-exact to the oracle, tests/test_gpu_matching.py) followed by ph_ratio_filter_l2 —
keep must be equal bit for bit, idx equal where keep is set and -1 where it is not."""
import os

import numpy as np
import pytest
import torch

from pose_estimation_amd import matching, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["own_norms", "given_norms"])
def norms_mode(request, monkeypatch):
    """Every test of this module runs twice: with the row norms computed inside the matcher call (fused into the passes
    that read the tables: NULL norms at the C ABI) and with ph_row_sqnorm's norms handed in (the two-pass path)."""
    monkeypatch.setattr(matching, "OWN_NORMS", request.param == "own_norms")


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _both(q, t, ratio, device, **kw):
    idx, d1, d2 = matching.match_l2(q, t, **kw)
    keep = matching.ratio_filter_l2(idx, d1, d2, ratio)
    fi, fk = matching.match_l2_ratio(q, t, ratio, **kw)
    idx, keep = idx.cpu().numpy(), keep.cpu().numpy()
    idx = np.where(keep == 1, idx, -1)                      # the fused call reports no match for a rejected query
    return idx, keep, fi.cpu().numpy(), fk.cpu().numpy(), d1.cpu().numpy(), d2.cpu().numpy()


@pytest.mark.parametrize("ratio", [0.5, 0.75, 0.9, 1.0])
def test_fused_equals_two_calls_on_sift_like_pairs(device, ratio):
    scs = [synthetic.two_view_pair(n=2048, seed=40 + p) for p in range(3)]
    q = _t(np.stack([s["desc1"] for s in scs]), device)
    t = _t(np.stack([s["desc2"] for s in scs]), device)
    idx, keep, fi, fk, d1, d2 = _both(q, t, ratio, device)
    assert np.array_equal(fk, keep)
    assert np.array_equal(fi, idx)
    assert ratio in (0.5, 1.0) or 0 < keep.sum() < keep.size


def test_fused_full_size_and_keep_fraction(device):
    sc = synthetic.two_view_pair(n=8192, seed=1002)
    q, t = _t(sc["desc1"][None], device), _t(sc["desc2"][None], device)
    idx, keep, fi, fk, _, _ = _both(q, t, 0.75, device)
    assert np.array_equal(fk, keep) and np.array_equal(fi, idx)
    assert 0.4 < keep.mean() < 0.8


_MORE = int(os.environ.get("POSEHOT_FUZZ_SEEDS", "0"))      # extra seeds for a one-off soak run


@pytest.mark.parametrize("seed", range(10 + _MORE))
def test_fused_fuzz(device, seed):
    """Random shapes, ragged counts, shared view tables, duplicates (exact ties), near ties, signed data, scales."""
    rng = np.random.default_rng(9100 + seed)
    V = int(rng.integers(1, 4)); n = int(rng.choice([1, 2, 3, 33, 64, 65, 257, 700, 1500]))
    B = int(rng.integers(1, 6))
    tab = rng.standard_normal((V, n, 128)).astype(np.float32)
    kind = seed % 5
    if kind == 0:
        tab = np.abs(tab)
    if kind == 1 and n > 4:                                   # duplicates and near-duplicates: ties and ratio ~ 1
        tab[:, n // 2:] = tab[:, : n - n // 2] + (rng.standard_normal((V, n - n // 2, 128)) * 1e-4).astype(np.float32)
        tab[:, -1] = tab[:, 0]
    if kind == 2:
        tab *= np.float32(10.0 ** rng.integers(-6, 6))
    if kind == 3:                                             # rows of very different norms
        tab *= (10.0 ** rng.uniform(-3, 3, (V, n, 1))).astype(np.float32)
    if kind == 4 and n > 2:                                   # clustered: many scores inside the error band
        c = rng.standard_normal((V, 4, 128)).astype(np.float32)
        tab = (c[np.arange(V)[:, None], rng.integers(0, 4, (V, n))] + 1e-3 * tab).astype(np.float32)
    tab = np.ascontiguousarray(tab, np.float32)
    vq = rng.integers(0, V, B).astype(np.int32); vt = rng.integers(0, V, B).astype(np.int32)
    n_q = rng.integers(0, n + 1, B).astype(np.int32); n_t = rng.integers(0, n + 1, B).astype(np.int32)
    n_q[0] = n; n_t[0] = n
    d = _t(tab, device)
    ratio = float(rng.choice([0.6, 0.75, 0.8, 0.95, 1.0]))
    kw = dict(view_q=_t(vq, device), view_t=_t(vt, device), n_q=_t(n_q, device), n_t=_t(n_t, device))
    idx, keep, fi, fk, _, _ = _both(d, d, ratio, device, **kw)
    for b in range(B):
        assert np.array_equal(fi[b, :n_q[b]], idx[b, :n_q[b]]), (b, n, kind)
        assert np.array_equal(fk[b, :n_q[b]], keep[b, :n_q[b]]), (b, n, kind)
        assert (fi[b, n_q[b]:] == -1).all() and (fk[b, n_q[b]:] == 0).all()


def test_fused_ratio_boundary_cases(device):
    """Queries built so that d1 / d2 sits at, just below and just above the ratio: the interval test must hand every one of
    them to the exact path and the result must still be that of the two calls."""
    rng = np.random.default_rng(5)
    n = 512
    t = rng.standard_normal((n, 128)).astype(np.float32)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    q = np.zeros((n, 128), np.float32)
    for i in range(n):
        a, b = t[i], t[(i + 1) % n]
        # a point on the segment a..b whose distances to a and b have a ratio near 0.75 (+- a few 1e-7 .. 1e-2)
        lam = 0.75 / 1.75 + rng.choice([0.0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3, 1e-2, -1e-2])
        q[i] = a + lam * (b - a)
    idx, keep, fi, fk, d1, d2 = _both(_t(q[None], device), _t(t[None], device), 0.75, device)
    assert np.array_equal(fk, keep) and np.array_equal(fi, idx)
    r = np.sqrt(d1[0] / np.maximum(d2[0], 1e-30))
    assert (np.abs(r - 0.75) < 0.02).mean() > 0.5          # the construction did put the ratios on the boundary
    assert 0.1 < keep.mean() < 0.9


def test_fused_empty_and_single_train_row(device):
    q = np.random.default_rng(1).standard_normal((1, 10, 128)).astype(np.float32)
    idx, keep = matching.match_l2_ratio(_t(q, device), _t(np.zeros((1, 0, 128), np.float32), device), 0.75)
    assert (idx.cpu().numpy() == -1).all() and (keep.cpu().numpy() == 0).all()
    t = q[:, :1].copy()
    idx, keep = matching.match_l2_ratio(_t(q, device), _t(t, device), 0.75)
    assert (idx.cpu().numpy() == -1).all() and (keep.cpu().numpy() == 0).all()  # no second neighbour: rejected, no match


def test_fused_nan_rows_nothing_faults_and_clean_queries_agree(device):
    """posehot.h: descriptors must be finite.  With a NaN train row and a NaN query the launch still completes, and the
    queries whose clean top-2 did not involve the NaN row get the keep / idx of the clean two-call path."""
    rng = np.random.default_rng(5)
    q = synthetic.sift_like(rng, 700, 128); t = synthetic.sift_like(rng, 900, 128)
    idx0, keep0, _, _, _, _ = _both(_t(q[None], device), _t(t[None], device), 0.8, device)
    j0, i0 = 123, 45
    t2, q2 = t.copy(), q.copy()
    t2[j0, 17] = np.nan; q2[i0, 3] = np.nan
    fi, fk = (x.cpu().numpy()[0] for x in matching.match_l2_ratio(_t(q2[None], device), _t(t2[None], device), 0.8))
    d = ((q[:, None, :].astype(np.float64) - t[None, :, :].astype(np.float64)) ** 2).sum(-1)
    order = np.argsort(d, axis=1)
    same = (np.arange(700) != i0) & (order[:, 0] != j0) & (order[:, 1] != j0)
    assert same.sum() > 600
    assert np.array_equal(fk[same], keep0[0][same]) and np.array_equal(fi[same], idx0[0][same])
    assert not (fi == j0).any()
