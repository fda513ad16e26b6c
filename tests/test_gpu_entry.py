"""Entry points on the synthetic board ring (SURVEY.md §8d cfg 1): formats and the acceptance grade."""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_entry_points_stdout_equals_golden_text(device, tmp_path):
    """a-7 byte compatibility: the whole stdout of both entry points on the synthetic 8-camera detections file equals
    tests/golden/{main,validate}_stdout.txt — texts written from the reference's print formats with numbers from the
    CPU path (tests/golden/make_golden.py::stdout_texts), independent of the GPU library."""
    sys.path.insert(0, ROOT)
    import main as main_mod
    import validate_pose as val_mod
    from pose_estimation_amd import entry, synthetic
    from tests.golden.make_golden import STDOUT_SEED
    det_file, html = str(tmp_path / "dets.npz"), str(tmp_path / "scene" / "camera_poses_3d.html")
    entry.save_detections(det_file, synthetic.board_views(8, seed=STDOUT_SEED))
    buf = io.StringIO()
    with redirect_stdout(buf):
        main_mod.main(["--detections", det_file, "--output-html", html])
    want = open(os.path.join(ROOT, "tests", "golden", "main_stdout.txt"), encoding="utf-8").read().replace("{HTML}", html)
    assert buf.getvalue() == want
    assert os.path.getsize(html) > 10000 and "Cam 8" in open(html, encoding="utf-8").read()
    buf = io.StringIO()
    with redirect_stdout(buf):
        val_mod.validate_all_images(detections=det_file)
    assert buf.getvalue() == open(os.path.join(ROOT, "tests", "golden", "validate_stdout.txt"), encoding="utf-8").read()


def test_main_and_validate_on_synthetic_ring(device, tmp_path):
    sys.path.insert(0, ROOT)
    import main as main_mod
    import validate_pose as val_mod
    from pose_estimation_amd import entry, synthetic
    views = synthetic.board_views(8)
    det_file = str(tmp_path / "dets.npz")
    entry.save_detections(det_file, views)
    buf = io.StringIO()
    with redirect_stdout(buf):
        results = main_mod.main(["--detections", det_file, "--output-html", str(tmp_path / "scene.html")])
    out = buf.getvalue()
    assert len(results) == 8 and set(results[0]) == {'filename', 'image_path', 'corners', 'ids', 'rvec', 'tvec',
                                                     'camera_position', 'camera_forward', 'camera_up', 'R_cam',
                                                     'quaternion', 'num_markers'}
    assert "📊 CAMERA POSE SUMMARY" in out and "  Cam 1 ↔ Cam 2: " in out and "synthetic_00.png" in out
    centre = np.array([0.1875, 0.1175, 0.0])
    for r, v in zip(results, views):                      # cameras sit on the 0.6 m ring, 0.4 m up
        C = r['camera_position']
        assert abs(np.linalg.norm((C - centre)[:2]) - 0.6) < 2e-3 and abs(C[2] - 0.4) < 2e-3
        assert np.allclose(C, -v["R"].T @ v["t"], atol=2e-3)
    buf = io.StringIO()
    with redirect_stdout(buf):
        errs = val_mod.validate_all_images(detections=det_file)
    out = buf.getvalue()
    assert len(errs) == 8 and np.mean(errs) < 1.0        # 0.3 px corner noise -> EXCELLENT band
    assert "EXCELLENT" in out and "📊 VALIDATION SUMMARY" in out and "Reprojection Error: mean=" in out
