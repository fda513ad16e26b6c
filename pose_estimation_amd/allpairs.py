"""All-pairs relative pose over resident views (BASELINE.json configs[2]/[3]; SURVEY.md §8d cfg 3/4, §8e).

The reference's only all-pairs loop is the O(n^2) camera-distance table of main.py:458-463; north_star generalises
it to "every pair of views -> relative pose", sharded over ranks by `sharding.run_pair_list` (pair_id % world, one
final gather).  `AllPairs.chunk` is the per-rank chunk processor handed to that runner.
"""
from __future__ import annotations

import numpy as np
import torch

from . import matching, pipeline


class AllPairs:
    """One rank's chunk processor of cfg 3/4: views resident, a chunk = `len(ids)` pairs through the relative-pose
    pipeline.  RANSAC samples are keyed by the GLOBAL pair id (and a fixed seed), so the row of a pair does not
    depend on the rank, chunk or batch slot that computes it — sharded and unsharded runs agree row for row."""

    SEED = 20240

    def __init__(self, desc, kp, norms, thr_sq, n_samples, f32, ratio=0.8, refine=True):
        self.desc, self.kp, self.norms, self.thr_sq, self.n_samples, self.f32 = desc, kp, norms, thr_sq, n_samples, f32
        self.ratio, self.refine = ratio, refine
        self.real_models = torch.zeros((), dtype=torch.int64, device=desc.device)

    def chunk(self, k, ids, vi, vj):
        dev = self.desc.device
        q = torch.as_tensor(np.ascontiguousarray(vi), dtype=torch.int32).to(dev, non_blocking=True)
        t = torch.as_tensor(np.ascontiguousarray(vj), dtype=torch.int32).to(dev, non_blocking=True)
        key = ids.to(torch.int32).to(dev, non_blocking=True)
        if self.f32:
            match = lambda: (*matching.match_l2_ratio(self.desc, self.desc, self.ratio, q, t, norm_q=self.norms,
                                                      norm_t=self.norms), "l2_keep")
        else:
            match = lambda: (*matching.match_hamming(self.desc, self.desc, q, t), "hamming")
        out = pipeline.relative_pose_batch(match, self.kp, self.kp, q, t, None, self.thr_sq, ratio=self.ratio,
                                           n_samples=self.n_samples, seed=self.SEED, item_key=key, want_counts=True,
                                           refine=self.refine)
        self.real_models += out["n_solutions"].sum()
        return out["table"]
