"""The C-ABI library loads and exports every symbol include/posehot.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "posehot.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ph_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 18
    lib = ctypes.CDLL(os.path.join(ROOT, "pose_estimation_amd", "libposehot.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    from pose_estimation_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.ph_version() == 100
    assert lib.ph_status_string(0) == b"ok" and b"workspace" in lib.ph_status_string(-3)


def test_argument_errors_without_gpu():
    """Status codes, never exceptions or aborts (the reference's sentinel-return convention)."""
    from pose_estimation_amd import _lib
    lib = _lib.load()
    assert lib.ph_match_hamming_top2(None, None, None, None, None, None, 1, 8, 8, 32, None, None, None, None, 0, None) == -1
    assert lib.ph_match_hamming_top2(None, None, None, None, None, None, 1, 8, 8, 64, None, None, None, None, 0, None) == -2
    assert lib.ph_match_hamming_top2(None, None, None, None, None, None, -1, 8, 8, 32, None, None, None, None, 0, None) == -1
    assert lib.ph_match_hamming_top2(None, None, None, None, None, None, 0, 8, 8, 32, None, None, None, None, 0, None) == 0
    assert lib.ph_match_l2_top2(None, None, None, None, None, None, None, None, 1, 8, 8, 96, 0, None, None, None, None, 0, None) == -2
    assert lib.ph_match_l2_top2(None, None, None, None, None, None, None, None, 1, 8, 8, 128, 2, None, None, None, None, 0, None) == -2
    assert lib.ph_match_l2_top2(None, None, None, None, None, None, None, None, 1, 8, 8, 128, 1, None, None, None, None, 0, None) == -1
    assert lib.ph_score_essential(None, None, None, None, None, None, 1, 4, 4, 1.0, None, None) == -1
    assert lib.ph_triangulate(None, None, None, None, None, None, None, None, 1, 1, 4, None, None, None) == -1
    assert lib.ph_select_best(None, 0, 4, None, None, None) == 0
    assert lib.ph_match_hamming_ws_bytes(1, 32768, 32768) > 0
    assert lib.ph_match_hamming_ws_bytes(64, 32768, 32768) >= 64 * 32768 * 128     # expanded train tile images
    assert lib.ph_match_l2_ws_bytes(16, 8192, 8192, 128, 0) == 0 and lib.ph_match_l2_ws_bytes(1, 8192, 8192, 128, 0) > 0
    assert lib.ph_match_l2_ws_bytes(16, 8192, 8192, 128, 1) >= 16 * 8192 * 2 * (32 * 8 + 4)


def test_every_entry_point_takes_empty_and_negative_sizes_without_touching_the_device():
    """All sizes 0 with NULL pointers: a status (ok, or invalid / unsupported for a zero dimension), no launch, no dereference
    (this process has no GPU: a launch or a HIP call would come back as PH_ERR_LAUNCH).  All sizes -1: PH_ERR_INVALID_ARG."""
    from pose_estimation_amd import _lib
    lib = _lib.load()
    called = 0
    for name, (res, args) in _lib.SIGNATURES.items():
        if res is not ctypes.c_int or not args:
            continue
        for val, allowed in ((0, (0, -1, -2)), (-1, (-1,))):
            a = [None if t is ctypes.c_void_p else 1.0 if t is ctypes.c_float else val for t in args]
            assert getattr(lib, name)(*a) in allowed, (name, val)
            called += 1
    assert called >= 2 * 40


def test_product_never_imports_oracle():
    """The product package must not reach into oracle/ (the checker is not the thing shipped)."""
    pkg = os.path.join(ROOT, "pose_estimation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower().replace("the cpu oracle", "").replace("the oracle restates", ""), f
    for f in ("main.py", "validate_pose.py"):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert "oracle" not in open(p).read()


def test_missing_tensor_on_cpu_fails_loudly():
    import pytest
    import torch
    from pose_estimation_amd import PosehotError, matching
    with pytest.raises(PosehotError):
        matching.match_hamming(torch.zeros((1, 4, 32), dtype=torch.uint8), torch.zeros((1, 4, 32), dtype=torch.uint8))
