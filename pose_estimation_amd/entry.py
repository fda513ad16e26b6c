"""Drivers behind the repo-root entry points main.py / validate_pose.py.

They keep the reference's printed line formats and result-record keys (SURVEY.md §8 rows a-7,
a-8; main.py:198-214, 439-463; validate_pose.py:374-375, 390-410) while batching the per-image
arithmetic: all images' poses come from one ph_refine_pose_lm launch and all reprojection
statistics from one ph_reprojection_error launch.

Detections.  The reference's front end (cv2.imread + ArucoDetector, main.py:168-178) is out of
scope and cv2 is not installed here, so detections come from, in this order:
  1. `--detections file.npz` (keys: filenames, ids_<k>, corners_<k>),
  2. the image folder through cv2, when cv2 imports and the folder holds images,
  3. the synthetic 8-camera ring around the reference's board (SURVEY.md §8d cfg 1).
"""
from __future__ import annotations

import glob
import os

import numpy as np
import torch

from . import board as bd
from . import synthetic


def load_detections(image_folder, detections_file=None, rotate_to_portrait=True):
    """-> (list of dict(filename, image_path, corners, ids), source string)."""
    if detections_file:
        z = np.load(detections_file, allow_pickle=False)
        names = [str(s) for s in z["filenames"]]
        dets = []
        for k, name in enumerate(names):
            c = z[f"corners_{k}"].astype(np.float32)
            dets.append(dict(filename=name, image_path=os.path.join(image_folder or "", name),
                             corners=tuple(c[i].reshape(1, 4, 2) for i in range(len(c))),
                             ids=z[f"ids_{k}"].astype(np.int32).reshape(-1, 1)))
        return dets, f"detections file {detections_file}"
    paths = set()
    if image_folder and os.path.isdir(image_folder):
        for ext in ("*.jpg", "*.jpeg", "*.png", "*.bmp", "*.tiff"):
            paths.update(glob.glob(os.path.join(image_folder, ext)))
    if paths:
        try:
            import cv2  # noqa: F401  (the reference's front end; absent in this image)
        except ImportError:
            cv2 = None
        if cv2 is not None:
            dictionary = cv2.aruco.getPredefinedDictionary(cv2.aruco.DICT_6X6_250)
            detector = cv2.aruco.ArucoDetector(dictionary, cv2.aruco.DetectorParameters())
            dets = []
            for p in sorted(paths):
                image = cv2.imread(p)
                if image is None:
                    continue
                if rotate_to_portrait and image.shape[1] > image.shape[0]:
                    image = cv2.rotate(image, cv2.ROTATE_90_COUNTERCLOCKWISE)
                gray = cv2.cvtColor(image, cv2.COLOR_BGR2GRAY) if image.ndim == 3 else image
                corners, ids, _ = detector.detectMarkers(gray)
                dets.append(dict(filename=os.path.basename(p), image_path=p, corners=corners, ids=ids))
            return dets, f"cv2 detection on {image_folder}"
    views = synthetic.board_views(8)
    return ([dict(filename=v["filename"], image_path=os.path.join(image_folder or "", v["filename"]),
                  corners=v["corners"], ids=v["ids"]) for v in views],
            "synthetic 8-camera ring (no cv2 / no images: SURVEY.md §8d cfg 1)")


def save_detections(path, dets):
    out = {"filenames": np.array([d["filename"] for d in dets])}
    for k, d in enumerate(dets):
        out[f"ids_{k}"] = np.asarray(d["ids"], np.int32).reshape(-1)
        out[f"corners_{k}"] = np.stack([np.asarray(c, np.float32).reshape(4, 2) for c in d["corners"]])
    np.savez_compressed(path, **out)


def process_detections(dets, board, camera_matrix, dist_coeffs, verbose=True):
    """The loop of main.py:163-214 on detections: result records with the reference's 12 keys."""
    results = []
    # all images' poses in one Levenberg-Marquardt launch; the loop below only formats
    poses = bd.estimate_board_poses([(d["corners"], d["ids"]) for d in dets], board, camera_matrix, dist_coeffs)
    # ... and all camera inversions + quaternions in one ph_camera_from_pose launch (main.py:112-143, batched)
    solved = [k for k, (rvec, _, _) in enumerate(poses) if rvec is not None]
    cams = {}
    if solved:
        rt = np.stack([np.concatenate([np.asarray(poses[k][0], np.float64).reshape(3),
                                       np.asarray(poses[k][1], np.float64).reshape(3)]) for k in solved])
        cam = {name: v.cpu().numpy() for name, v in
               bd.camera_from_pose_batched(torch.from_numpy(rt).to(torch.device("cuda", torch.cuda.current_device()))).items()}
        cams = {k: tuple(cam[name][i] for name in ("position", "forward", "up", "R_cam_to_world", "quaternion"))
                for i, k in enumerate(solved)}
    for k, (d, (rvec, tvec, num_markers)) in enumerate(zip(dets, poses)):
        filename, corners, ids = d["filename"], d["corners"], d["ids"]
        if verbose:
            print(f"\n🔍 Processing: {filename}")
        if ids is None:
            if verbose:
                print("  ⚠️ No markers detected")
            continue
        if verbose:
            print(f"  ✓ Detected {len(ids)} markers: {np.asarray(ids).flatten().tolist()}")
        if rvec is None:
            if verbose:
                print("  ⚠️ Could not estimate pose")
            continue
        cam_pos, cam_forward, cam_up, R_cam, quaternion = cams[k]
        if verbose:
            print(f"  ✓ Camera position: [{cam_pos[0]:.3f}, {cam_pos[1]:.3f}, {cam_pos[2]:.3f}] m")
            print(f"  ✓ Quaternion (w,x,y,z): [{quaternion[0]:.4f}, {quaternion[1]:.4f}, {quaternion[2]:.4f}, {quaternion[3]:.4f}]")
        results.append({
            'filename': filename, 'image_path': d.get("image_path", filename), 'corners': corners, 'ids': ids,
            'rvec': rvec, 'tvec': tvec, 'camera_position': cam_pos, 'camera_forward': cam_forward,
            'camera_up': cam_up, 'R_cam': R_cam, 'quaternion': quaternion, 'num_markers': num_markers})
    return results


def format_pose_summary(results):
    """The text print_pose_summary (main.py:439-463) prints, as a list of lines."""
    L = ["\n" + "=" * 100, "📊 CAMERA POSE SUMMARY", "=" * 100,
         f"{'Image':<25} {'#':>4} {'X(m)':>8} {'Y(m)':>8} {'Z(m)':>8} │ {'qw':>7} {'qx':>7} {'qy':>7} {'qz':>7}", "-" * 100]
    for r in results:
        pos, q = r['camera_position'], r['quaternion']
        L.append(f"{r['filename']:<25} {r['num_markers']:>4} {pos[0]:>8.3f} {pos[1]:>8.3f} {pos[2]:>8.3f} │ "
                 f"{q[0]:>7.4f} {q[1]:>7.4f} {q[2]:>7.4f} {q[3]:>7.4f}")
    L += ["-" * 100, "\n📐 RELATIVE DISTANCES BETWEEN CAMERAS (meters):", "-" * 80]
    for i in range(len(results)):
        for j in range(i + 1, len(results)):
            distance = np.linalg.norm(results[i]['camera_position'] - results[j]['camera_position'])
            L.append(f"  Cam {i+1} ↔ Cam {j+1}: {distance:.3f} m")
    return L


def batched_reprojection_stats(results, board, camera_matrix, dist_coeffs):
    """calculate_reprojection_error (validate_pose.py:260-288) for all images in ONE launch."""
    if not results:
        return []
    N = 4 * len(board.getIds())
    B = len(results)
    obj = np.zeros((B, N, 3), np.float32); img = np.zeros((B, N, 2), np.float32)
    pose = np.zeros((B, 6), np.float64); nv = np.zeros(B, np.int32)
    for b, r in enumerate(results):
        o, i = board.matchImagePoints(r['corners'], r['ids'])
        n = 0 if o is None else len(o)
        nv[b] = n
        if n:
            obj[b, :n], img[b, :n] = o.reshape(-1, 3), i.reshape(-1, 2)
        pose[b, :3], pose[b, 3:] = np.asarray(r['rvec']).reshape(3), np.asarray(r['tvec']).reshape(3)
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a: torch.from_numpy(a).to(dev)
    _, stats = bd.reprojection_error_batched(t(obj), t(img), t(pose), t(nv), camera_matrix, dist_coeffs)
    stats = stats.cpu().numpy()
    return [None if nv[b] == 0 else {'mean': stats[b, 0], 'std': stats[b, 1], 'max': stats[b, 2], 'min': stats[b, 3],
                                     'num_points': int(stats[b, 4])} for b in range(B)]


def grade_line(mean_error):
    """validate_pose.py:400-407."""
    if mean_error < 1.0:
        return "  ✅ EXCELLENT: Mean error < 1 pixel - Pose estimation is very accurate!"
    if mean_error < 2.0:
        return "  ✅ GOOD: Mean error < 2 pixels - Pose estimation is accurate."
    if mean_error < 5.0:
        return "  ⚠️ FAIR: Mean error < 5 pixels - Pose estimation is acceptable."
    return "  ❌ POOR: Mean error >= 5 pixels - Check calibration or marker detection."
