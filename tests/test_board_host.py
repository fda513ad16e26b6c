"""Host-side mirror of the reference's board functions (no GPU needed): shapes, sentinels, conventions."""
import json
import os

import numpy as np

from pose_estimation_amd import board, geometry, synthetic

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_constants_match_reference():
    assert (board.MARKERS_X, board.MARKERS_Y) == (11, 7)
    assert board.MARKER_LENGTH_MM == 25.0 and board.MARKER_SEPARATION_MM == 10.0
    assert board.CAMERA_MATRIX.dtype == np.float64 and board.CAMERA_MATRIX[0, 0] == 3108.95 and board.CAMERA_MATRIX[1, 2] == 1940.89
    assert board.DIST_COEFFS.tolist() == [0.148, -0.510, 0.003, -0.0004, 0.635]
    assert board.ROTATE_LANDSCAPE_TO_PORTRAIT is True and board.IMAGE_FOLDER == "test_data/ChArUco"


def test_match_image_points_shapes_and_sentinels():
    _, b = board.create_aruco_board()
    v = synthetic.board_views(1, seed=1, drop=0.3)[0]
    obj, img = b.matchImagePoints(v["corners"], v["ids"])
    n = len(v["ids"])
    assert obj.shape == (4 * n, 1, 3) and img.shape == (4 * n, 1, 2) and obj.dtype == np.float32 and img.dtype == np.float32
    assert np.array_equal(obj[:4, 0], b.getObjPoints()[int(v["ids"][0, 0])])
    assert b.matchImagePoints((), None) == (None, None)
    assert b.matchImagePoints(v["corners"][:1], np.array([[999]], np.int32)) == (None, None)
    assert board.estimate_board_pose(v["corners"], None, b, board.CAMERA_MATRIX, board.DIST_COEFFS) == (None, None, 0)
    assert board.estimate_board_pose((), np.zeros((0, 1), np.int32), b, board.CAMERA_MATRIX, board.DIST_COEFFS) == (None, None, 0)
    r = board.estimate_board_pose(v["corners"][:1], np.array([[999]], np.int32), b, board.CAMERA_MATRIX, board.DIST_COEFFS)
    assert r == (None, None, 1)


def test_camera_inversion_and_quaternion_order():
    d = json.load(open(os.path.join(G, "quaternions.json")))
    for c in d["cases"][:20]:
        Rcw = np.array(c["R"]); q = np.array(c["quat_xyzw"])
        rvec = board.rotation_to_rvec(Rcw.T)
        tvec = np.array([0.1, -0.2, 0.7])
        pos, fwd, up, Rm, quat = board.get_camera_position_and_orientation(rvec, tvec)
        assert np.allclose(Rm, Rcw, atol=1e-9)
        assert np.allclose(pos, -Rcw @ tvec) and np.allclose(fwd, Rcw[:, 2]) and np.allclose(up, -Rcw[:, 1])
        assert np.allclose(quat, [q[3], q[0], q[1], q[2]], atol=1e-9) or np.allclose(quat, [-q[3], -q[0], -q[1], -q[2]], atol=1e-9)
        assert np.allclose(board._quat_xyzw(Rcw), q, atol=1e-12)
    assert pos.shape == (3,) and Rm.shape == (3, 3) and quat.shape == (4,)


def test_board_3d_points_follow_reference_quirks():
    pts = board.create_board_3d_points(0.375, 0.235)
    assert pts["boundary"].shape == (5, 3) and pts["boundary"].dtype == np.float32
    assert len(pts["vertical_lines"]) == 12 and len(pts["horizontal_lines"]) == 8
    assert np.isclose(pts["vertical_lines"][3][0][0], 3 * 0.035) and pts["vertical_lines"][11][0][0] == 0.375
    assert pts["marker_centers"].shape == (77, 3) and np.allclose(pts["marker_centers"][0], [0.0125, 0.0125, 0])
    assert len(pts["marker_boxes"]) == 77 and pts["marker_boxes"][0].shape == (5, 3)


def test_decompose_essential_contains_truth():
    R = synthetic.rot_y(np.deg2rad(-10.0)); t = -R @ np.array([0.5, 0.0, 0.0])
    Rc, tc = geometry.decompose_essential(synthetic.essential_from_pose(R, t)[None])
    errs = [geometry.rotation_angle_deg(Rc[0, c], R) + geometry.direction_angle_deg(tc[0, c], t) for c in range(4)]
    assert min(errs) < 1e-5 and sorted(errs)[1] > 1.0
    for c in range(4):
        assert np.isclose(np.linalg.det(Rc[0, c]), 1.0)
