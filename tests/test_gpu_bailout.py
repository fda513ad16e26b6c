"""Bail-out scoring (ph_score_*_bailout): the winner — index, count, ties to the lowest index — is that of the
exhaustive counts (oracle), every reported count is a lower bound of the exact one, and the maximum is exact."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from pose_estimation_amd import ransac, synthetic

pytestmark = pytest.mark.gpu


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _check(cnt, ref):
    assert (cnt <= ref).all()
    for b in range(ref.shape[0]):
        bi, bc = orc.select_best(ref[b])
        assert cnt[b].max() == bc and int(cnt[b].argmax()) == bi          # numpy argmax = first maximum = lowest index
        assert np.array_equal(cnt[b][ref[b] == bc], ref[b][ref[b] == bc])   # every hypothesis at the maximum is exact


@pytest.mark.parametrize("B,H,N", [(3, 500, 3000), (8, 4100, 5000), (1, 64, 700), (40, 33, 8192)])
def test_essential_bailout_winner_is_exhaustive_winner(device, B, H, N):
    rng = np.random.default_rng(B * 1000 + H)
    E = np.zeros((B, H, 9), np.float32)
    pts = [np.zeros((B, N), np.float32) for _ in range(4)]
    n_valid = rng.integers(N // 2, N + 1, B).astype(np.int32)
    n_valid[0] = N
    for b in range(B):
        sc = synthetic.two_view_pair(n=N, seed=900 + b)
        pr = sc["partner"]; ok = pr >= 0
        x1, y1 = sc["kp1"][:, 0], sc["kp1"][:, 1]
        x2, y2 = x1.copy(), y1.copy()
        x2[ok], y2[ok] = sc["kp2"][pr[ok], 0], sc["kp2"][pr[ok], 1]
        for a, s in zip(pts, (x1, y1, x2, y2)):
            a[b] = s
        Et = synthetic.essential_from_pose(sc["R"], sc["t"]); Et /= np.linalg.norm(Et)
        E[b] = rng.standard_normal((H, 9)) * 0.3
        E[b] /= np.linalg.norm(E[b], axis=1, keepdims=True)
        near = Et.reshape(9) + rng.standard_normal((H // 4, 9)) * 0.002       # a quarter of the hypotheses are decent
        E[b, rng.choice(H, H // 4, replace=False)] = near / np.linalg.norm(near, axis=1, keepdims=True)
        E[b, H // 2] = Et.reshape(9); E[b, H - 1] = Et.reshape(9)               # the best one twice: tie -> lowest index
        E[b, 5:9] = 0.0                                                         # empty solver slots
    thr = np.float32((1.0 / synthetic.FX) ** 2)
    d = [_t(a, device) for a in pts]
    cnt = ransac.score_essential(_t(E, device), *d, float(thr), _t(n_valid, device), bailout=True).cpu().numpy()
    full = ransac.score_essential(_t(E, device), *d, float(thr), _t(n_valid, device)).cpu().numpy()
    ref = np.stack([orc.score_essential(E[b], *(a[b][:n_valid[b]] for a in pts), thr) for b in range(B)])
    assert np.array_equal(full, ref)
    _check(cnt, ref)
    bi, bc = ransac.select_best(_t(cnt, device))
    bi2, bc2 = ransac.select_best(_t(full, device))
    assert torch.equal(bi, bi2) and torch.equal(bc, bc2)
    if B * ((H + 31) // 32) >= 1024 or N <= 1024:       # one point split: the bail-out really ran
        assert (cnt < ref).any()


def test_pnp_bailout_winner_is_exhaustive_winner(device):
    B, H, N = 4, 2048, 20000
    fr = synthetic.pnp_frames(B, N, outliers=0.3, noise_px=1.0, seed=77)
    rng = np.random.default_rng(77)
    K = np.array([[synthetic.FX, 0, synthetic.CX], [0, synthetic.FY, synthetic.CY], [0, 0, 1.0]])
    P = np.zeros((B, H, 12), np.float32)
    for b in range(B):
        base = synthetic.perturbed_poses(rng, fr["R"][b], fr["t"][b], 32, rot_sigma=0.01, t_sigma=0.02)
        rep = base[rng.integers(1, 32, H)]
        rep[:, :, 3] += rng.normal(0, 0.02, (H, 3))
        rep[H // 3] = base[0]; rep[H - 2] = base[0]
        P[b] = (K @ rep).reshape(H, 12).astype(np.float32)
        P[b, 100:300] = 0.0
    pts = [fr[k] for k in "XYZuv"]
    d = [_t(a, device) for a in pts]
    thr = np.float32(64.0)
    cnt = ransac.score_pnp(_t(P, device), *d, float(thr), bailout=True).cpu().numpy()
    ref = np.stack([orc.score_pnp(P[b], *(a[b] for a in pts), thr) for b in range(B)])
    _check(cnt, ref)
    assert all(int(cnt[b].argmax()) == H // 3 for b in range(B))
