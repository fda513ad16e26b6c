"""pose_estimation_amd — MI355X-native camera-pose hot path.

Host side in Python (the reference is Python: /root/reference/main.py, validate_pose.py);
compute in hand-written gfx950 HIP kernels behind the C ABI of include/posehot.h.
"""
from ._lib import PosehotError, load, LIB_PATH  # noqa: F401

__all__ = ["PosehotError", "load", "LIB_PATH"]
