"""A relative-pose batch of fixed shape as ONE replayable hipGraph (BASELINE.json configs[1]: "single pair").

`pipeline.relative_pose_batch` is a chain of about thirty launches, each through a Python wrapper that allocates its
outputs: at 64 pairs per batch that host work (0.23 ms) hides behind the GPU, at one pair it IS the latency.  A
`RelativePosePlan` runs the same chain once under stream capture — every output and workspace then comes from the graph's
private pool, sized once — and afterwards a batch costs one `hipGraphLaunch`: no allocation, no Python per stage, no
launch gaps wider than the hardware's.

What varies between replays lives in device tensors the caller fills (`view_q`, `view_t`: which views of the resident
tables form the pairs; `key`: the Philox key of every pair); table pointers, shapes, thresholds and the Philox seed are
baked in.  The replayed kernels are the ones the eager pipeline launches, with the same arguments, so a replay's pose
table is bit-identical to `pipeline.relative_pose_batch` for the same tables, views, seed and keys
(tests/test_gpu_plan.py).  The 5-point solver's polynomial records sit in the library's static device scratch
(`ph_solve_essential_5pt_noalloc`, include/posehot.h) when the batch fits its 8192-sample limit.

No reference counterpart: the reference processes one image per loop trip on the host (main.py:163-214).
"""
from __future__ import annotations

import torch

from . import matching, pipeline


class RelativePosePlan:
    """desc_q / desc_t: resident descriptor tables ``[V, n, 128]`` float32 (L2) or ``[V, n, 32]`` uint8 (Hamming);
    kp_q / kp_t: ``[V, n, 2]`` normalised keypoints; B pairs per replay.

    fresh_norms=True lets the matcher compute the row norms of the pairs' views inside every replay (a front end that hands over
    new views every time — what bench.py's headline step does); False computes them once, here (views resident across calls)."""

    def __init__(self, desc_q, desc_t, kp_q, kp_t, B, thr_sq, ratio=0.75, n_samples=920, seed=0, refine=True,
                 fresh_norms=True, warmup=2):
        if not desc_q.is_cuda:
            raise ValueError("a plan binds device tables (there is no CPU path)")
        self.dev = desc_q.device
        self.B, self.thr_sq, self.ratio, self.n_samples, self.seed, self.refine = int(B), float(thr_sq), float(ratio), int(n_samples), int(seed), bool(refine)
        self.desc_q, self.desc_t, self.kp_q, self.kp_t = desc_q, desc_t, kp_q, kp_t
        self.float_desc = desc_q.dtype == torch.float32
        self.fresh_norms = bool(fresh_norms) and self.float_desc
        self.view_q = torch.arange(self.B, dtype=torch.int32, device=self.dev) % desc_q.shape[0]
        self.view_t = torch.arange(self.B, dtype=torch.int32, device=self.dev) % desc_t.shape[0]
        self.key = torch.arange(self.B, dtype=torch.int32, device=self.dev)
        self.norm_q = self.norm_t = None
        if self.float_desc and not self.fresh_norms:
            self.norm_q = matching.row_sqnorm(desc_q)
            self.norm_t = self.norm_q if desc_t is desc_q else matching.row_sqnorm(desc_t)
        self.noalloc = self.B * self.n_samples <= 8192
        self.stream = torch.cuda.Stream(device=self.dev)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.stream):
            for _ in range(max(1, warmup)):             # first-call work (library load, the scratch symbol's address) outside the capture
                self._chain()
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.out = self._chain()
        self.table = self.out["table"]

    def _chain(self):
        if self.float_desc:
            def match():
                nq, nt = (None, None) if self.fresh_norms else (self.norm_q, self.norm_t)     # None: computed inside the matcher call
                return (*matching.match_l2_ratio(self.desc_q, self.desc_t, self.ratio, self.view_q, self.view_t,
                                                 norm_q=nq, norm_t=nt), "l2_keep")
        else:
            def match():
                return (*matching.match_hamming(self.desc_q, self.desc_t, self.view_q, self.view_t), "hamming")
        return pipeline.relative_pose_batch(match, self.kp_q, self.kp_t, self.view_q, self.view_t, None, self.thr_sq,
                                            ratio=self.ratio, n_samples=self.n_samples, seed=self.seed, item_key=self.key,
                                            want_counts=True, refine=self.refine, noalloc=self.noalloc)

    def set_pairs(self, view_q=None, view_t=None, key=None):
        """Fill the plan's pair list / Philox keys (device copies on the current stream; replays see them in stream order)."""
        for dst, src in ((self.view_q, view_q), (self.view_t, view_t), (self.key, key)):
            if src is not None:
                dst.copy_(torch.as_tensor(src, dtype=torch.int32), non_blocking=True)

    def replay(self):
        """One batch.  Returns the plan's static result dict: ``table`` [B,16] and the pipeline's other outputs, overwritten
        by the next replay.  Replays of one plan share its buffers: do not overlap them on different streams."""
        self.graph.replay()
        return self.out
