"""RelativePosePlan (pose_estimation_amd/plan.py): the relative-pose chain captured once as a hipGraph.

A replay launches the kernels the eager pipeline launches, with the same arguments, so its pose table must equal
`pipeline.relative_pose_batch` BIT FOR BIT for the same tables, views, seed and Philox keys — at one pair per replay
(BASELINE.json configs[1]) and at eight, after the pair list and the keys were changed between replays, and for the
Hamming matcher.  Parity of the eager pipeline with the CPU oracle: tests/test_gpu_pipeline.py."""
import numpy as np
import pytest
import torch

from pose_estimation_amd import matching, pipeline, plan, synthetic

pytestmark = pytest.mark.gpu


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _eager(dq, dt, kq, kt, vq, vt, key, thr_sq, S, seed, ratio=0.75):
    match = lambda: (*matching.match_l2_ratio(dq, dt, ratio, vq, vt), "l2_keep")
    return pipeline.relative_pose_batch(match, kq, kt, vq, vt, None, thr_sq, ratio=ratio, n_samples=S, seed=seed,
                                        item_key=key, want_counts=True)


@pytest.mark.parametrize("B", [1, 8])
def test_graph_replay_equals_eager_pipeline_bit_for_bit(device, B):
    n, S, seed, V = 2048, 96, 31, 4
    scs = [synthetic.two_view_pair(n=n, seed=500 + v) for v in range(V)]
    dq = _t(np.stack([s["desc1"] for s in scs]), device); dt = _t(np.stack([s["desc2"] for s in scs]), device)
    kq = _t(np.stack([s["kp1"] for s in scs]), device); kt = _t(np.stack([s["kp2"] for s in scs]), device)
    thr_sq = float(np.float32((1.0 / synthetic.FX) ** 2))
    p = plan.RelativePosePlan(dq, dt, kq, kt, B, thr_sq, n_samples=S, seed=seed)
    assert p.noalloc                                   # B * S <= 8192: the solver's records sit in the static scratch
    rng = np.random.default_rng(B)
    for trip in range(3):                              # new pair list and keys before every replay
        v = rng.integers(0, V, B).astype(np.int32)
        key = rng.integers(0, 1 << 30, B).astype(np.int32)
        p.set_pairs(v, v, key)
        out = p.replay()
        torch.cuda.synchronize()
        got = out["table"].cpu().numpy().copy()
        want = _eager(dq, dt, kq, kt, _t(v, device), _t(v, device), _t(key, device), thr_sq, S, seed)
        assert np.array_equal(got.view(np.uint32), want["table"].cpu().numpy().view(np.uint32))
        assert torch.equal(out["n_match"], want["n_match"]) and torch.equal(out["mask"], want["mask"])
        assert (got[:, 15] == pipeline.STATUS_OK).all() and (got[:, 12] > 100).all()


def test_graph_replay_hamming(device):
    sc = synthetic.binary_views(3, 4096, seed=8)
    d, kp = _t(sc["desc"], device), _t(sc["kp"], device)
    thr_sq = float(np.float32((1.5 / synthetic.FX) ** 2))
    p = plan.RelativePosePlan(d, d, kp, kp, 2, thr_sq, ratio=0.8, n_samples=64, seed=5)
    vq, vt, key = np.array([0, 1], np.int32), np.array([1, 2], np.int32), np.array([10, 11], np.int32)
    p.set_pairs(vq, vt, key)
    got = p.replay()["table"].cpu().numpy().copy()
    q, t, k = _t(vq, device), _t(vt, device), _t(key, device)
    want = pipeline.relative_pose_batch(lambda: (*matching.match_hamming(d, d, q, t), "hamming"), kp, kp, q, t, None, thr_sq,
                                        ratio=0.8, n_samples=64, seed=5, item_key=k)
    assert np.array_equal(got.view(np.uint32), want["table"].cpu().numpy().view(np.uint32))
    assert (got[:, 15] == pipeline.STATUS_OK).all()
