"""The headless 3-D scene writer (/root/reference/main.py:233-432 without fig.show()): scene
contents as data, layout arithmetic, and the HTML file — CPU only (host NumPy/SciPy pose functions, no GPU call)."""
import os

import numpy as np

from pose_estimation_amd import board, scene, synthetic


def _records(n=10):
    out = []
    for k, v in enumerate(synthetic.board_views(n, seed=3)):
        rvec = board.rotation_to_rvec(v["R"])
        pos, fwd, up, Rcw, q = board.get_camera_position_and_orientation(rvec, v["t"])
        out.append({'filename': v["filename"], 'image_path': v["filename"], 'corners': v["corners"], 'ids': v["ids"],
                    'rvec': rvec.reshape(3, 1), 'tvec': v["t"].reshape(3, 1), 'camera_position': pos, 'camera_forward': fwd,
                    'camera_up': up, 'R_cam': Rcw, 'quaternion': q, 'num_markers': len(v["ids"])})
    return out


def test_scene_contents_follow_the_reference():
    res = _records(10)
    tr = scene.scene_traces(res, 0.375, 0.235)
    assert len(tr) == 2 + 4 * len(res) + 3
    assert tr[0]['name'] == 'ArUco Board' and tr[0]['x'] == [0, 0.375, 0.375, 0, 0] and tr[0]['line'] == dict(color='blue', width=5)
    assert tr[1]['type'] == 'mesh3d' and tr[1]['opacity'] == 0.3 and tr[1]['i'] == [0, 0] and tr[1]['k'] == [2, 3]
    for n, r in enumerate(res):
        m, fw, cone, up = tr[2 + 4 * n: 6 + 4 * n]
        c = scene.CAMERA_COLORS[n % 8]                                   # eight colours, cycled (main.py:278-287)
        assert m['marker'] == dict(size=10, color=c, symbol='diamond') and m['text'] == [f"Cam {n+1}"]
        assert m['name'] == f"{r['filename']} ({r['num_markers']} markers)"
        tip = r['camera_position'] + 0.15 * r['camera_forward']
        assert np.allclose([fw['x'][1], fw['y'][1], fw['z'][1]], tip) and fw['showlegend'] is False
        assert cone['type'] == 'cone' and np.allclose([cone['x'][0], cone['y'][0], cone['z'][0]], tip) and cone['sizeref'] == 0.03
        assert np.allclose([cone['u'][0], cone['v'][0], cone['w'][0]], r['camera_forward'])
        assert np.allclose([up['x'][1], up['y'][1], up['z'][1]], r['camera_position'] + 0.08 * r['camera_up'])
        assert up['line']['dash'] == 'dash'
    assert [t['name'] for t in tr[-3:]] == ['X-axis', 'Y-axis', 'Z-axis'] and tr[-1]['z'] == [0.0, 0.15]
    lay = scene.scene_layout(res, 0.375, 0.235)
    P = np.array([r['camera_position'] for r in res])
    half = max(np.ptp(P[:, 0]), np.ptp(P[:, 1]), np.ptp(P[:, 2]), 0.375, 0.235) * 0.6 + 0.2
    cx = (P[:, 0].mean() + 0.375 / 2) / 2
    assert np.allclose(lay['scene']['xaxis']['range'], [cx - half, cx + half]) and lay['scene']['aspectmode'] == 'cube'
    assert np.allclose(lay['scene']['zaxis']['range'], [P[:, 2].mean() / 2 - half, P[:, 2].mean() / 2 + half])
    assert lay['scene']['camera']['eye'] == dict(x=1.5, y=1.5, z=1.5) and lay['title']['text'] == scene.TITLE


def test_scene_html_is_written_without_showing(tmp_path, capsys):
    res = _records(3)
    path = str(tmp_path / "sub" / "camera_poses_3d.html")
    spec = scene.write_scene_html(res, 0.375, 0.235, path)
    assert capsys.readouterr().out == f"\n✅ Interactive 3D visualization saved to: {path}\n"       # main.py:431
    html = open(path, encoding="utf-8").read()
    assert len(spec['data']) == 17 and "Cam 3" in html and "Board Surface" in html and "plotly" in html.lower()
    assert scene.write_scene_html([], 0.375, 0.235, str(tmp_path / "none.html")) is None
    assert not os.path.exists(tmp_path / "none.html")
    assert "No valid poses to visualize" in capsys.readouterr().out
