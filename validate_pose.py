#!/usr/bin/env python3
"""Pose validation by reprojection error — entry point with the reference's name, flags, constants
and printed formats (/root/reference/validate_pose.py), running on pose_estimation_amd (MI355X).

The reprojection statistics of all images come from one batched GPU launch
(ph_reprojection_error); overlay drawing (validate_pose.py:187-257) is out of scope.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from pose_estimation_amd import board as bd  # noqa: E402
from pose_estimation_amd import entry  # noqa: E402


def _resolve_path(script_dir, path_value, default_rel):
    if path_value is None:
        return os.path.join(script_dir, default_rel)
    return path_value if os.path.isabs(path_value) else os.path.join(script_dir, path_value)


def validate_all_images(image_folder=None, output_folder=None, rotate_to_portrait=bd.ROTATE_LANDSCAPE_TO_PORTRAIT,
                        detections=None):
    print("=" * 80)
    print("🔍 Pose Estimation Validation")
    print("=" * 80)
    dictionary, board = bd.create_aruco_board()
    board_width_m = (bd.MARKERS_X * bd.MARKER_LENGTH_MM + (bd.MARKERS_X - 1) * bd.MARKER_SEPARATION_MM) / 1000.0
    board_height_m = (bd.MARKERS_Y * bd.MARKER_LENGTH_MM + (bd.MARKERS_Y - 1) * bd.MARKER_SEPARATION_MM) / 1000.0
    print(f"\n📋 Board: {board_width_m*1000:.1f} × {board_height_m*1000:.1f} mm")
    script_dir = os.path.dirname(os.path.abspath(__file__))
    image_folder = _resolve_path(script_dir, image_folder, bd.IMAGE_FOLDER)
    output_folder = _resolve_path(script_dir, output_folder, bd.OUTPUT_FOLDER)
    dets, source = entry.load_detections(image_folder, detections, rotate_to_portrait)
    if not dets:
        print(f"❌ No images found in {image_folder}")
        return None
    print(f"[detections: {source}]", file=sys.stderr)
    print(f"📷 Found {len(dets)} images\n")

    results = entry.process_detections(dets, board, bd.CAMERA_MATRIX, bd.DIST_COEFFS, verbose=False)
    by_name = {r['filename']: r for r in results}
    stats = dict(zip([r['filename'] for r in results],
                     entry.batched_reprojection_stats(results, board, bd.CAMERA_MATRIX, bd.DIST_COEFFS)))
    all_errors = []
    for d in dets:
        print(f"🔍 Validating: {d['filename']}")
        if d['ids'] is None:
            print("  ⚠️ No markers detected")
            continue
        print(f"  ✓ Detected {len(d['ids'])} markers")
        if d['filename'] not in by_name:
            print("  ⚠️ Could not estimate pose")
            continue
        error_stats = stats[d['filename']]
        if error_stats:
            print(f"  📊 Reprojection Error: mean={error_stats['mean']:.2f}px, "
                  f"max={error_stats['max']:.2f}px, std={error_stats['std']:.2f}px")
            all_errors.append(error_stats['mean'])
    if all_errors:
        print("\n" + "=" * 80)
        print("📊 VALIDATION SUMMARY")
        print("=" * 80)
        print(f"  Mean reprojection error: {np.mean(all_errors):.2f} pixels")
        print(f"  Max reprojection error:  {np.max(all_errors):.2f} pixels")
        print(f"  Min reprojection error:  {np.min(all_errors):.2f} pixels")
        print(f"  Std deviation:           {np.std(all_errors):.2f} pixels")
        print("-" * 80)
        print(entry.grade_line(np.mean(all_errors)))
    return all_errors


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Validate ArUco board pose by reprojection error.")
    parser.add_argument("--image-folder", type=str, default=None, help="Path to input images (default: test_data/ChArUco)")
    parser.add_argument("--output-folder", type=str, default=None, help="Path to output images (default: test_data/pose_validation)")
    parser.add_argument("--no-rotate", action="store_true", help="Disable landscape-to-portrait rotation")
    parser.add_argument("--detections", type=str, default=None, help="npz of marker detections instead of images")
    args = parser.parse_args()
    validate_all_images(image_folder=args.image_folder, output_folder=args.output_folder,
                        rotate_to_portrait=not args.no_rotate, detections=args.detections)
