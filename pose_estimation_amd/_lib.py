"""ctypes binding of libposehot.so (C ABI: include/posehot.h).

PyTorch is plumbing here: it owns device memory and streams; every compute call goes through
the C ABI into hand-written gfx950 kernels.  There is no CPU fallback: if the shared library
is missing, importing a compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POSEHOT_LIB") or os.path.join(_HERE, "libposehot.so")   # POSEHOT_LIB: another build of the same ABI (experiments)

PH_OK = 0
PH_NO_INDEX = -1
PH_HAMMING_NONE = 0x7FFFFFFF
PRECISION_F32 = 0
PRECISION_F16_RERANK = 1
BAILOUT_SLOTS = 2048        # int32 words of scratch per item for ph_score_*_bailout (posehot.h)


class PosehotError(RuntimeError):
    """A libposehot entry point returned a negative status."""


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/posehot.h declaration by declaration.
SIGNATURES = {
    "ph_version": (_i, []),
    "ph_status_string": (C.c_char_p, [_i]),
    "ph_match_hamming_ws_bytes": (_sz, [_i, _i, _i]),
    "ph_match_hamming_top2": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "ph_hamming_view_images_bytes": (_sz, [_i, _i]),
    "ph_hamming_expand_views": (_i, [_p, _i, _i, _i, _p, _sz, _p]),
    "ph_match_hamming_images_ws_bytes": (_sz, [_i, _i, _i]),
    "ph_match_hamming_top2_images": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "ph_match_l2_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ph_row_sqnorm": (_i, [_p, _i64, _i, _p, _p]),
    "ph_match_l2_top2": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "ph_match_l2_top2_timed": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p, _p, _p]),
    "ph_match_l2_ratio": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _sz, _p, _p, _p]),
    "ph_ratio_filter_hamming": (_i, [_p, _p, _p, _i64, _f, _p, _p]),
    "ph_ratio_filter_l2": (_i, [_p, _p, _p, _i64, _f, _p, _p]),
    "ph_compact_matches": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "ph_sample_minimal": (_i, [C.c_uint32, _p, _i, _i, _i, _i, _p, _p]),
    "ph_sample_minimal_keyed": (_i, [C.c_uint32, _p, _p, _i, _i, _i, _i, _p, _p]),
    "ph_solve_essential_5pt_ws_bytes": (_sz, [_i, _i]),
    "ph_solve_essential_5pt": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "ph_solve_essential_5pt_noalloc": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "ph_gather_models": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "ph_compact_models": (_i, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "ph_compact_models_counted": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "ph_lookup_index": (_i, [_p, _p, _i, _i, _p, _p]),
    "ph_decompose_essential": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "ph_pose_table": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _p, _p, _p]),
    "ph_refine_relative_pose": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _p, _i, _p, _p, _p, _p]),
    "ph_solve_p3p": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "ph_score_essential": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p]),
    "ph_score_pnp": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p]),
    "ph_score_essential_bailout": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p]),
    "ph_score_pnp_bailout": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p]),
    "ph_select_best": (_i, [_p, _i, _i, _p, _p, _p]),
    "ph_select_best_mapped": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "ph_inlier_mask_essential": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p]),
    "ph_inlier_mask_pnp": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p]),
    "ph_triangulate": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "ph_project_points": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "ph_reprojection_error": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "ph_refine_pose_lm": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _p, _p, _p, _p]),
    "ph_refine_pose_lm_soa": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _p, _p, _p, _p]),
    "ph_refine_pose_lm_soa_ws_bytes": (_sz, [_i, _i]),
    "ph_refine_pose_lm_soa_ws": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _p, _p, _p, _p, _sz, _p]),
    "ph_pose_from_projection": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "ph_initial_board_pose": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _p, _p]),
    "ph_camera_from_pose": (_i, [_p, _i, _p, _p]),
    "ph_undistort_points": (_i, [_p, _p, _p, _i64, _i, _p, _p]),
}

_lib = None


def load() -> C.CDLL:
    """Load libposehot.so once; fail loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PosehotError(
            f"{LIB_PATH} not found: build it with `make -C pose_estimation_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of step
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != PH_OK:
        msg = load().ph_status_string(status).decode()
        raise PosehotError(f"{what}: {msg} (status {status})")


def ptr(t: torch.Tensor | None):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    """Validate a device operand: on the GPU, right dtype, contiguous."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise PosehotError(f"{name}: tensor must live on the GPU (there is no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def opt_i32(t, B: int, name: str):
    if t is None:
        return None
    t = dev(t, torch.int32, name)
    if t.numel() != B:
        raise ValueError(f"{name}: expected {B} elements, got {t.numel()}")
    return t


def workspace(nbytes: int, device) -> torch.Tensor | None:
    if nbytes == 0:
        return None
    return torch.empty(nbytes, dtype=torch.uint8, device=device)
