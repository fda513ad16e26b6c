"""GPU parity of the matchers against the CPU oracle, through the C ABI.

Bars: Hamming — bit-exact idx/d1/d2; L2 — bit-exact against the f32 fma-chain oracle, and
>= 99.9 % match-set agreement against the independent float64 sum of squared differences
(north_star), with the disagreeing rows required to be float near-ties.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from pose_estimation_amd import matching, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["own_norms", "given_norms"])
def norms_mode(request, monkeypatch):
    """Every test of this module runs twice: with the row norms computed inside the matcher call (fused into the passes
    that read the tables: NULL norms at the C ABI) and with ph_row_sqnorm's norms handed in (the two-pass path)."""
    monkeypatch.setattr(matching, "OWN_NORMS", request.param == "own_norms")


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (7, 5), (64, 64), (300, 1000), (1000, 300), (2048, 4096), (513, 129)])
def test_hamming_parity_random(device, nq, nt):
    rng = np.random.default_rng(nq * 7919 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    idx, d1, d2 = matching.match_hamming(_t(q[None], device), _t(t[None], device))
    ri, r1, r2 = orc.match_hamming_top2(q, t)
    assert np.array_equal(idx.cpu().numpy()[0], ri)
    assert np.array_equal(d1.cpu().numpy()[0], r1)
    assert np.array_equal(d2.cpu().numpy()[0], r2)


def test_hamming_ties_lowest_index(device):
    """Planted exact duplicates in the train set: the lowest index must win, d2 == d1."""
    rng = np.random.default_rng(5)
    t = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    t[650] = t[3]; t[200] = t[3]; t[699] = t[100]
    q = t[[3, 100, 650, 5]].copy()
    q[3, 0] ^= 1
    idx, d1, d2 = matching.match_hamming(_t(q[None], device), _t(t[None], device))
    ri, r1, r2 = orc.match_hamming_top2(q, t)
    assert idx.cpu().numpy()[0].tolist() == [3, 100, 3, 5] == ri.tolist()
    assert np.array_equal(d1.cpu().numpy()[0], r1) and np.array_equal(d2.cpu().numpy()[0], r2)
    assert d2.cpu().numpy()[0][0] == 0


def test_hamming_batched_views_and_ragged(device):
    """Pair list over a shared view table, with per-pair valid counts."""
    sc = synthetic.binary_views(5, 600, seed=11)
    vq = np.array([0, 0, 1, 3, 4], np.int32); vt = np.array([1, 2, 2, 4, 0], np.int32)
    n_q = np.array([600, 599, 1, 300, 0], np.int32); n_t = np.array([600, 2, 600, 1, 600], np.int32)
    d = _t(sc["desc"], device)
    idx, d1, d2 = matching.match_hamming(d, d, _t(vq, device), _t(vt, device), _t(n_q, device), _t(n_t, device))
    idx, d1, d2 = idx.cpu().numpy(), d1.cpu().numpy(), d2.cpu().numpy()
    for b in range(5):
        ri, r1, r2 = orc.match_hamming_top2(sc["desc"][vq[b], :n_q[b]], sc["desc"][vt[b], :n_t[b]])
        assert np.array_equal(idx[b, :n_q[b]], ri) and np.array_equal(d1[b, :n_q[b]], r1) and np.array_equal(d2[b, :n_q[b]], r2)
        assert (idx[b, n_q[b]:] == -1).all()
    # single train row: no second neighbour -> sentinel, ratio test rejects
    assert (d2[3, :300] == orc.HAMMING_NONE).all()
    keep = matching.ratio_filter_hamming(_t(idx, device), _t(d1, device), _t(d2, device), 0.8).cpu().numpy()
    assert np.array_equal(keep, orc.ratio_filter_hamming(idx, d1, d2, 0.8))
    assert keep[3].sum() == 0 and keep[4].sum() == 0


def test_hamming_far_rows_beat_padding(device):
    """Rows beyond the valid count are zero nibbles in the tile image (score 0 = distance 128): every real row here is
    FARTHER than that (distance 200-256), in the staged tile the valid rows end in and with a ragged count."""
    rng = np.random.default_rng(77)
    nt, nq = 200, 70
    t = np.zeros((nt, 32), np.uint8)
    t[rng.random((nt, 32)) < 0.05] = 1                    # a few set bits
    q = np.full((nq, 32), 255, np.uint8)
    q[rng.random((nq, 32)) < 0.05] = 254
    for n_t in (1, 63, 65, 130, 200):
        idx, d1, d2 = matching.match_hamming(_t(q[None], device), _t(t[None], device), n_t=_t(np.array([n_t], np.int32), device))
        ri, r1, r2 = orc.match_hamming_top2(q, t[:n_t])
        assert r1.min() > 128
        assert np.array_equal(idx.cpu().numpy()[0], ri) and np.array_equal(d1.cpu().numpy()[0], r1)
        assert np.array_equal(d2.cpu().numpy()[0], r2)


def test_hamming_empty_train(device):
    q = np.zeros((1, 10, 32), np.uint8)
    t = np.zeros((1, 0, 32), np.uint8)
    idx, d1, d2 = matching.match_hamming(_t(q, device), _t(t, device))
    assert (idx.cpu().numpy() == -1).all() and (d1.cpu().numpy() == orc.HAMMING_NONE).all()


def test_hamming_planted_matches_recovered(device):
    """Domain property at a larger size: bit-flipped copies match their landmark partner."""
    sc = synthetic.binary_views(2, 8192, seed=3)
    d = _t(sc["desc"], device)
    idx, d1, d2 = matching.match_hamming(d[0:1], d[1:2])
    idx = idx.cpu().numpy()[0]
    pos = {int(l): j for j, l in enumerate(sc["ids"][1])}
    want = np.array([pos.get(int(l), -1) for l in sc["ids"][0]])
    shared = want >= 0
    assert shared.sum() > 1000
    assert (idx[shared] == want[shared]).mean() > 0.999
    ri, r1, r2 = orc.match_hamming_top2(sc["desc"][0], sc["desc"][1])
    assert np.array_equal(idx, ri) and np.array_equal(d1.cpu().numpy()[0], r1) and np.array_equal(d2.cpu().numpy()[0], r2)


PRECISIONS = [0, 1]     # f32 MFMA kernel / fp16 candidate filter + exact re-rank: same bit-exact contract


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("nq,nt,dim", [(1, 1, 128), (5, 3, 128), (64, 64, 128), (300, 1000, 128), (1000, 257, 128),
                                        (2048, 2048, 128), (333, 555, 64)])
def test_l2_parity_bit_exact(device, nq, nt, dim, precision):
    rng = np.random.default_rng(nq * 31 + nt + dim)
    q = synthetic.sift_like(rng, nq, dim)
    t = synthetic.sift_like(rng, nt, dim)
    idx, d1, d2 = matching.match_l2(_t(q[None], device), _t(t[None], device), precision=precision)
    ri, r1, r2 = orc.match_l2_top2(q, t)
    assert np.array_equal(idx.cpu().numpy()[0], ri)
    assert np.array_equal(d1.cpu().numpy()[0].view(np.uint32), r1.view(np.uint32))
    assert np.array_equal(d2.cpu().numpy()[0].view(np.uint32), r2.view(np.uint32))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_l2_signed_data_and_duplicates(device, precision):
    """General (signed) data and exact duplicate train rows: ties -> lowest index, d1 == d2."""
    rng = np.random.default_rng(17)
    t = rng.standard_normal((500, 128)).astype(np.float32)
    t[400] = t[7]; t[450] = t[7]
    q = np.concatenate([t[[7, 400]], rng.standard_normal((30, 128)).astype(np.float32)])
    idx, d1, d2 = matching.match_l2(_t(q[None], device), _t(t[None], device), precision=precision)
    ri, r1, r2 = orc.match_l2_top2(q, t)
    assert np.array_equal(idx.cpu().numpy()[0], ri) and idx.cpu().numpy()[0][0] == 7 and idx.cpu().numpy()[0][1] == 7
    assert np.array_equal(d1.cpu().numpy()[0].view(np.uint32), r1.view(np.uint32))
    assert np.array_equal(d2.cpu().numpy()[0].view(np.uint32), r2.view(np.uint32))


def test_l2_match_set_agreement_with_float64(device):
    """north_star: >= 99.9 % agreement with an independent float64 matcher; disagreements are near-ties."""
    sc = synthetic.two_view_pair(n=4096, seed=21)
    idx, d1, d2 = matching.match_l2(_t(sc["desc1"][None], device), _t(sc["desc2"][None], device))
    idx = idx.cpu().numpy()[0]
    ri, r1, r2 = orc.match_l2_top2_f64(sc["desc1"], sc["desc2"])
    agree = idx == ri
    assert agree.mean() >= 0.999
    bad = np.nonzero(~agree)[0]
    if bad.size:  # documented tie case: best and runner-up closer than f32 resolution of the dot product
        assert np.all((r2[bad] - r1[bad]) < 1e-5)
    keep = matching.ratio_filter_l2(_t(idx[None], device), d1, d2, 0.75).cpu().numpy()[0]
    keep64 = (r1 < 0.75 ** 2 * r2)
    assert (keep.astype(bool) == keep64).mean() >= 0.999
    # planted partners are recovered
    pl = sc["partner"] >= 0
    assert (idx[pl] == sc["partner"][pl]).mean() > 0.99


def _assert_same_bits(a, b):
    for x, y in zip(a, b):
        x, y = x.cpu().numpy(), y.cpu().numpy()
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_l2_filter_adversarial_inputs(device):
    """The candidate filter against inputs built to break it; the f32 kernel (oracle-checked above) is the
    reference at these sizes.  (a) train rows sorted by increasing similarity: every row beats the running
    bound, the lists overflow and the exact fallback scan must take over; (b) all train rows identical: every
    row ties; (c) wide dynamic range: row norms from 1e-3 to 1e3, the error band is set by the largest norm;
    (d) near-duplicates differing in the last bits: ordering decided far below the filter's resolution."""
    rng = np.random.default_rng(99)
    q = synthetic.sift_like(rng, 700, 128)
    # (a)
    t = synthetic.sift_like(rng, 3000, 128)
    order = np.argsort((t @ q[0]) - 0.5 * (t * t).sum(1))
    ta = t[order]
    # (b)
    tb = np.repeat(t[:1], 1500, axis=0)
    # (c)
    tc = (t[:2000] * np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (2000, 1)))).astype(np.float32)
    qc = (q * np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (700, 1)))).astype(np.float32)
    # (d)
    base = synthetic.sift_like(rng, 40, 128)
    td = np.repeat(base, 50, axis=0)
    td = (td.view(np.uint32) + rng.integers(0, 4, td.shape, dtype=np.uint32)).view(np.float32)
    for qq, tt in ((q, ta), (q, tb), (qc, tc), (np.concatenate([base, q[:100]]), td)):
        a = matching.match_l2(_t(qq[None], device), _t(tt[None], device), precision=0)
        f = matching.match_l2(_t(qq[None], device), _t(tt[None], device), precision=1)
        _assert_same_bits(a, f)
    ri, r1, r2 = orc.match_l2_top2(q, ta)
    f = matching.match_l2(_t(q[None], device), _t(ta[None], device), precision=1)
    assert np.array_equal(f[0].cpu().numpy()[0], ri)


def test_l2_filter_fp16_range_cases(device):
    """The filter computes in fp16 after a per-pair power-of-two scaling.  Cases aimed at that: (e) row norms spread
    over 12 decades on both sides — most elements land below fp16's normal range or at zero, the band's absolute term
    must cover them; (f) integer-valued 0..255 descriptors (OpenCV SIFT convention), norms ~500; (g) one giant row
    among unit rows (it sets the scale, everything else is tiny); (h) everything scaled by 1e-16: the scales leave
    the supported range and every query takes the exact path; (i) all-zero train rows / zero queries; (j) large
    magnitudes (1e15), where unscaled fp16 would overflow."""
    rng = np.random.default_rng(1234)
    q = synthetic.sift_like(rng, 300, 128)
    t = synthetic.sift_like(rng, 2500, 128)
    cases = []
    te = (t * np.exp(rng.uniform(np.log(1e-8), np.log(1e4), (2500, 1)))).astype(np.float32)
    qe = (q * np.exp(rng.uniform(np.log(1e-8), np.log(1e4), (300, 1)))).astype(np.float32)
    cases.append((qe, te))
    cases.append((np.floor(q * 512).clip(0, 255).astype(np.float32), np.floor(t * 512).clip(0, 255).astype(np.float32)))
    tg = t.copy(); tg[1234] *= 3.0e4
    cases.append((q, tg))
    cases.append(((q * 1e-16).astype(np.float32), (t * 1e-16).astype(np.float32)))
    tz = t.copy(); tz[:] = 0.0
    qz = q.copy(); qz[::3] = 0.0
    cases.append((q, tz)); cases.append((qz, t))
    cases.append(((q * 1e15).astype(np.float32), (t * 1e15).astype(np.float32)))
    for qq, tt in cases:
        a = matching.match_l2(_t(qq[None], device), _t(tt[None], device), precision=0)
        f = matching.match_l2(_t(qq[None], device), _t(tt[None], device), precision=1)
        _assert_same_bits(a, f)
    # and once against the CPU oracle, on the widest-range case
    ri, r1, r2 = orc.match_l2_top2(qe, te)
    f = matching.match_l2(_t(qe[None], device), _t(te[None], device), precision=1)
    assert np.array_equal(f[0].cpu().numpy()[0], ri)
    assert np.array_equal(f[1].cpu().numpy()[0].view(np.uint32), r1.view(np.uint32))


def test_l2_filter_full_size_equals_f32_kernel(device):
    """BASELINE cfg 2 shape, 16 pairs of 8192 x 8192 x 128 (no train split): both algorithms, same bits."""
    rng = np.random.default_rng(5)
    V = 6
    tab = np.stack([synthetic.sift_like(rng, 8192, 128) for _ in range(V)])
    tab[1, :4000] = synthetic.perturb_sift(rng, tab[0, :4000], 0.05)
    vq = np.arange(16, dtype=np.int32) % V; vt = (np.arange(16, dtype=np.int32) * 5 + 1) % V
    d = _t(tab, device)
    a = matching.match_l2(d, d, _t(vq, device), _t(vt, device), precision=0)
    f = matching.match_l2(d, d, _t(vq, device), _t(vt, device), precision=1)
    _assert_same_bits(a, f)
    one = matching.match_l2(d[:1], d[1:2], precision=1)           # single pair: train rows split over blocks
    a1 = matching.match_l2(d[:1], d[1:2], precision=0)
    _assert_same_bits(a1, one)
    assert (one[0][0, :4000].cpu().numpy() == np.arange(4000)).mean() > 0.99


@pytest.mark.parametrize("precision", PRECISIONS)
def test_l2_batched_ragged(device, precision):
    rng = np.random.default_rng(2)
    tab = np.stack([synthetic.sift_like(rng, 400, 128) for _ in range(4)])
    vq = np.array([0, 1, 2], np.int32); vt = np.array([3, 0, 1], np.int32)
    n_q = np.array([400, 77, 0], np.int32); n_t = np.array([1, 400, 400], np.int32)
    d = _t(tab, device)
    idx, d1, d2 = matching.match_l2(d, d, _t(vq, device), _t(vt, device), _t(n_q, device), _t(n_t, device),
                                    precision=precision)
    idx, d1, d2 = idx.cpu().numpy(), d1.cpu().numpy(), d2.cpu().numpy()
    for b in range(3):
        ri, r1, r2 = orc.match_l2_top2(tab[vq[b], :n_q[b]], tab[vt[b], :n_t[b]])
        assert np.array_equal(idx[b, :n_q[b]], ri)
        assert np.array_equal(d1[b, :n_q[b]].view(np.uint32), r1.view(np.uint32))
        assert np.array_equal(d2[b, :n_q[b]].view(np.uint32), r2.view(np.uint32))
        assert (idx[b, n_q[b]:] == -1).all()
    assert np.isinf(d2[0, :400]).all()


@pytest.mark.parametrize("nq,nt", [(1000, 900), (1, 5), (1024, 1024), (1025, 300), (3000, 64), (8192, 8192), (5000, 7)])
def test_compact_matches(device, nq, nt):
    """Stable compaction against the oracle: sizes below, at and across the 1024-query pieces a pair is cut into (a block per
    piece counts the kept matches in front of it), indices outside [0, nt) and PH_NO_INDEX dropped, a pair without a kept
    match, a pair that keeps everything."""
    rng = np.random.default_rng(9 + nq)
    B = 4
    idx = rng.integers(0, nt, (B, nq)).astype(np.int32)
    keep = (rng.random((B, nq)) < 0.4).astype(np.int32)
    keep[2] = 0
    keep[3] = 1
    bad = rng.random((B, nq)) < 0.05
    idx[bad] = np.where(rng.random(int(bad.sum())) < 0.5, -1, nt + 3)      # never a match, whatever keep says
    kq = rng.standard_normal((B, nq, 2)).astype(np.float32)
    kt = rng.standard_normal((B, nt, 2)).astype(np.float32)
    out = matching.compact_matches(_t(idx, device), _t(keep, device), _t(kq, device), _t(kt, device))
    for b in range(B):
        valid = (idx[b] >= 0) & (idx[b] < nt)                 # the contract drops such rows; the oracle takes them as not kept
        ref = orc.compact_matches(np.where(valid, idx[b], 0), keep[b] * valid, kq[b], kt[b])
        n = ref["n_match"]
        assert int(out["n_match"][b]) == n
        for k in ("x1", "y1", "x2", "y2", "match_q", "match_t"):
            assert np.array_equal(out[k][b, :n].cpu().numpy(), ref[k][:n])


@pytest.mark.parametrize("rows,dim", [(1, 128), (63, 128), (1000, 128), (4097, 64), (300, 36)])
def test_row_sqnorm_bit_exact(device, rows, dim):
    """Natural-order fma chain, bit for bit; dims that are a multiple of 32 take the coalesced kernel."""
    rng = np.random.default_rng(rows + dim)
    a = rng.standard_normal((rows, dim)).astype(np.float32)
    got = matching.row_sqnorm(_t(a[None], device)).cpu().numpy()[0]
    assert np.array_equal(got.view(np.uint32), orc.row_sqnorm(a).view(np.uint32))


@pytest.mark.parametrize("precision", [0, 1])
def test_l2_empty_train(device, precision):
    """A [V, 0, 128] train table: every query gets PH_NO_INDEX / +inf, for both algorithms (the Hamming twin of this
    case is test_hamming_empty_train)."""
    from pose_estimation_amd import _lib as L
    rng = np.random.default_rng(0)
    q = torch.from_numpy(synthetic.sift_like(rng, 70, 128)[None]).to(device)
    t = torch.empty((1, 0, 128), dtype=torch.float32, device=device)
    idx, d1, d2 = matching.match_l2(q, t, precision=precision)
    assert (idx == L.PH_NO_INDEX).all() and torch.isinf(d1).all() and torch.isinf(d2).all()
    keep = matching.ratio_filter_l2(idx, d1, d2, 0.75)
    assert int(keep.sum()) == 0


@pytest.mark.parametrize("precision", [0, 1])
def test_l2_nan_rows_are_never_preferred_and_nothing_faults(device, precision):
    """posehot.h: descriptors must be finite; a NaN row makes only the neighbours it touches unspecified — the launch
    completes, clean queries never pick the NaN train row, and those whose clean top-2 did not involve it keep their result."""
    rng = np.random.default_rng(5)
    q = synthetic.sift_like(rng, 700, 128); t = synthetic.sift_like(rng, 900, 128)
    tq, tt = torch.from_numpy(q[None]).to(device), torch.from_numpy(t[None]).to(device)
    idx0, a0, b0 = (x.cpu().numpy()[0] for x in matching.match_l2(tq, tt, precision=precision))
    ri, ra, rb = orc.match_l2_top2(q, t)
    assert np.array_equal(idx0, ri)
    j0, i0 = 123, 45
    t2, q2 = t.copy(), q.copy()
    t2[j0, 17] = np.nan; q2[i0, 3] = np.nan
    idx, a, b = (x.cpu().numpy()[0] for x in matching.match_l2(torch.from_numpy(q2[None]).to(device),
                                                               torch.from_numpy(t2[None]).to(device), precision=precision))
    clean = np.arange(700) != i0
    assert not (idx[clean] == j0).any()
    # second-best of the clean run (exact, by the float64 matcher) must not be j0 either for the result to be comparable
    d = ((q[:, None, :].astype(np.float64) - t[None, :, :].astype(np.float64)) ** 2).sum(-1)
    second = np.argsort(d, axis=1)[:, 1]
    same = clean & (idx0 != j0) & (second != j0)
    assert same.sum() > 600 and np.array_equal(idx[same], idx0[same]) and np.array_equal(a[same], a0[same])


def test_l2_own_norms_when_the_provisional_scales_do_not_stand(device):
    """Norms computed inside the call (NULL at the C ABI): the tile images and the query fragments are first scaled by a
    power of two taken from ROW 0 of either view; inputs whose largest norm that scale does not fit must take the fix-up
    pass (train) or the block's own scale with re-scaled presets (queries) — and give the f32 kernel's bits, as always.
    (a) train row 0 a thousand times smaller than the rest; (b) train row 0 all zero; (c) train row 0 huge; (d) query row 0
    tiny, the second query block (rows 256..511) a hundred times larger than the first, norms inside the third block
    spread over six decades; (e) both at once, two pairs sharing the tables through view lists, ragged valid counts."""
    rng = np.random.default_rng(2024)
    q = synthetic.sift_like(rng, 900, 128); t = synthetic.sift_like(rng, 2500, 128)
    ta = t.copy(); ta[0] *= 1e-3
    tb = t.copy(); tb[0] = 0.0
    tc = t.copy(); tc[0] *= 3e3
    qd = q.copy(); qd[0] *= 1e-4; qd[256:512] *= 100.0
    qd[512:768] *= np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (256, 1))).astype(np.float32)
    saved = matching.OWN_NORMS
    try:
        for qq, tt in ((q, ta), (q, tb), (q, tc), (qd, t), (qd, ta)):
            matching.OWN_NORMS = False
            want = matching.match_l2(_t(qq[None], device), _t(tt[None], device), precision=0)
            matching.OWN_NORMS = True
            got = matching.match_l2(_t(qq[None], device), _t(tt[None], device), precision=1)
            _assert_same_bits(want, got)
            idx, keep = matching.match_l2_ratio(_t(qq[None], device), _t(tt[None], device), 0.8)
            k0 = matching.ratio_filter_l2(*want, 0.8)
            assert torch.equal(keep, k0) and torch.equal(idx[keep.bool()], want[0][keep.bool()])
        ri, r1, r2 = orc.match_l2_top2(qd, ta)
        assert np.array_equal(got[0].cpu().numpy()[0], ri) and np.array_equal(got[1].cpu().numpy()[0].view(np.uint32), r1.view(np.uint32))
        # (e) view lists + ragged counts
        tq = _t(np.stack([qd, q]), device); tt = _t(np.stack([ta, tc]), device)
        vq = _t(np.array([0, 1, 0], np.int32), device); vt = _t(np.array([1, 0, 0], np.int32), device)
        nq_ = _t(np.array([900, 300, 777], np.int32), device); nt_ = _t(np.array([2500, 1000, 65], np.int32), device)
        matching.OWN_NORMS = False
        want = matching.match_l2(tq, tt, vq, vt, nq_, nt_, precision=0)
        matching.OWN_NORMS = True
        _assert_same_bits(want, matching.match_l2(tq, tt, vq, vt, nq_, nt_, precision=1))
    finally:
        matching.OWN_NORMS = saved


@pytest.mark.parametrize("V,n,B", [(3, 300, 5), (2, 64, 2), (4, 2049, 6), (1, 1, 1), (2, 8192, 1), (5, 1000, 64)])
def test_hamming_resident_view_images_equal_the_per_call_expansion(device, V, n, B):
    """ph_hamming_expand_views + ph_match_hamming_top2_images (train-side images of every view expanded once, read by view) give
    what ph_match_hamming_top2 gives, bit for bit: view lists, ragged valid counts (rows beyond a pair's n_t are in the resident
    image and must be masked), one pair with split train rows, many pairs."""
    rng = np.random.default_rng(50 + n + B)
    tab = _t(rng.integers(0, 256, (V, n, 32), dtype=np.uint8), device)
    vq = _t(rng.integers(0, V, B).astype(np.int32), device); vt = _t(rng.integers(0, V, B).astype(np.int32), device)
    n_q = rng.integers(0, n + 1, B).astype(np.int32); n_t = rng.integers(0, n + 1, B).astype(np.int32)
    n_q[0] = n; n_t[0] = n
    images = matching.hamming_view_images(tab)
    for nq_, nt_ in ((None, None), (_t(n_q, device), _t(n_t, device))):
        ref = matching.match_hamming(tab, tab, vq, vt, nq_, nt_)
        got = matching.match_hamming(tab, tab, vq, vt, nq_, nt_, images=images)
        for r, g in zip(ref, got):
            assert torch.equal(r, g)
    if B <= V:                                            # implicit view lists: pair b = view b of either table
        ref = matching.match_hamming(tab[:B], tab[:B])
        got = matching.match_hamming(tab[:B], tab[:B], images=matching.hamming_view_images(tab[:B]))
        for r, g in zip(ref, got):
            assert torch.equal(r, g)
