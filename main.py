#!/usr/bin/env python3
"""Camera pose from ArUco-board detections — entry point with the reference's name, constants and
printed formats (/root/reference/main.py), running on pose_estimation_amd (MI355X).

Differences, all forced by scope: detections come from cv2 only when it is
importable (their source is named on stderr); the 3-D scene (main.py:233-436) is written headless —
test_data/camera_poses_3d.html as the reference does, but nothing opens a browser (no fig.show(), main.py:434);
the per-image overlay files (main.py:217-228: cv2 drawing) are not written.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from pose_estimation_amd import board as bd  # noqa: E402
from pose_estimation_amd import entry  # noqa: E402
from pose_estimation_amd import scene  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description="ArUco board camera pose estimation (MI355X path).")
    ap.add_argument("--detections", default=None, help="npz of marker detections (see pose_estimation_amd/entry.py)")
    ap.add_argument("--image-folder", default=None, help="image folder (default: test_data/ChArUco, as the reference)")
    ap.add_argument("--output-html", default=None, help="3-D scene file (default: test_data/camera_poses_3d.html, as the reference)")
    args = ap.parse_args(argv)

    print("=" * 80)
    print("🎯 ArUco Board Camera Pose Estimation")
    print("=" * 80)
    dictionary, board = bd.create_aruco_board()
    board_width_m = (bd.MARKERS_X * bd.MARKER_LENGTH_MM + (bd.MARKERS_X - 1) * bd.MARKER_SEPARATION_MM) / 1000.0
    board_height_m = (bd.MARKERS_Y * bd.MARKER_LENGTH_MM + (bd.MARKERS_Y - 1) * bd.MARKER_SEPARATION_MM) / 1000.0
    print(f"\n📋 Board Configuration:")
    print(f"  - Grid: {bd.MARKERS_X} × {bd.MARKERS_Y} markers")
    print(f"  - Marker size: {bd.MARKER_LENGTH_MM} mm")
    print(f"  - Separation: {bd.MARKER_SEPARATION_MM} mm")
    print(f"  - Total size: {board_width_m*1000:.1f} × {board_height_m*1000:.1f} mm")
    print(f"  - Dictionary: DICT_6X6_250")

    script_dir = os.path.dirname(os.path.abspath(__file__))
    image_folder = args.image_folder or os.path.join(script_dir, bd.IMAGE_FOLDER)
    dets, source = entry.load_detections(image_folder, args.detections, bd.ROTATE_LANDSCAPE_TO_PORTRAIT)
    print(f"[detections: {source}]", file=sys.stderr)
    print(f"📷 Found {len(dets)} images")
    results = entry.process_detections(dets, board, bd.CAMERA_MATRIX, bd.DIST_COEFFS)
    if not results:
        print("\n❌ No valid poses could be estimated.")
        return []
    for line in entry.format_pose_summary(results):
        print(line)
    print("\n🎨 Creating interactive 3D visualization...")
    scene.write_scene_html(results, board_width_m, board_height_m,
                           args.output_html or os.path.join(script_dir, "test_data", "camera_poses_3d.html"))
    print("\n✅ Done!")
    return results


if __name__ == "__main__":
    main()
