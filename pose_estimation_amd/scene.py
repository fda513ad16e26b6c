"""Headless 3-D scene of the estimated camera poses.

The reference draws the board, one marker + viewing arrow + up arrow per camera and the world axes with Plotly,
writes `test_data/camera_poses_3d.html` and then calls `fig.show()`, which tries to open a browser
(/root/reference/main.py:233-436, :434).  This module produces the same scene from the same 12-key result records
(main.py:201-214) as DATA first — a list of trace dictionaries and a layout dictionary, testable without Plotly —
and writes the HTML with `plotly.graph_objects.Figure(...).write_html`; it never shows anything.
Scene contents and constants follow the reference: board outline blue / surface lightblue at 0.3 opacity
(main.py:253-273), eight camera colours cycled (main.py:278-281), 0.15 m viewing arrow with a 0.03 cone, 0.08 m dashed
up arrow (main.py:302-343), 0.15 m world axes (main.py:348-381), cube scene of half-width
0.6 * max(extent) + 0.2 around the mean of camera centroid and board centre (main.py:388-399).
"""
from __future__ import annotations

import json
import os

import numpy as np

CAMERA_COLORS = ('#FF6B6B', '#4ECDC4', '#45B7D1', '#96CEB4', '#FFEAA7', '#DDA0DD', '#98D8C8', '#F7DC6F')
FORWARD_LEN, UP_LEN, AXIS_LEN, CONE_SIZE = 0.15, 0.08, 0.15, 0.03
TITLE = '📷 Camera Pose Estimation - Interactive 3D Visualization'


def _segment(p, q, **style):
    p, q = np.asarray(p, float), np.asarray(q, float)
    return dict(type='scatter3d', x=[p[0], q[0]], y=[p[1], q[1]], z=[p[2], q[2]], **style)


def scene_traces(results, board_width_m, board_height_m):
    """Plotly trace dictionaries, in drawing order: board (2), four per camera, world axes (3)."""
    w, h = float(board_width_m), float(board_height_m)
    ring = np.array([[0, 0, 0], [w, 0, 0], [w, h, 0], [0, h, 0], [0, 0, 0]], float)
    traces = [
        dict(type='scatter3d', x=ring[:, 0].tolist(), y=ring[:, 1].tolist(), z=ring[:, 2].tolist(), mode='lines',
             line=dict(color='blue', width=5), name='ArUco Board'),
        dict(type='mesh3d', x=[0, w, w, 0], y=[0, 0, h, h], z=[0, 0, 0, 0], i=[0, 0], j=[1, 2], k=[2, 3],
             color='lightblue', opacity=0.3, name='Board Surface'),
    ]
    for n, r in enumerate(results):
        c = CAMERA_COLORS[n % len(CAMERA_COLORS)]
        pos = np.asarray(r['camera_position'], float)
        fwd, up = np.asarray(r['camera_forward'], float), np.asarray(r['camera_up'], float)
        tip = pos + FORWARD_LEN * fwd
        traces += [
            dict(type='scatter3d', x=[pos[0]], y=[pos[1]], z=[pos[2]], mode='markers+text',
                 marker=dict(size=10, color=c, symbol='diamond'), text=[f"Cam {n+1}"], textposition='top center',
                 name=f"{r['filename']} ({r['num_markers']} markers)"),
            _segment(pos, tip, mode='lines', line=dict(color=c, width=4), showlegend=False, hoverinfo='skip'),
            dict(type='cone', x=[tip[0]], y=[tip[1]], z=[tip[2]], u=[fwd[0]], v=[fwd[1]], w=[fwd[2]], sizemode='absolute',
                 sizeref=CONE_SIZE, colorscale=[[0, c], [1, c]], showscale=False, showlegend=False, hoverinfo='skip'),
            _segment(pos, pos + UP_LEN * up, mode='lines', line=dict(color=c, width=2, dash='dash'), showlegend=False,
                     hoverinfo='skip'),
        ]
    for axis, (label, colour) in enumerate((('X', 'red'), ('Y', 'green'), ('Z', 'blue'))):
        end = np.zeros(3)
        end[axis] = AXIS_LEN
        traces.append(_segment(np.zeros(3), end, mode='lines+text', line=dict(color=colour, width=3), text=['', label],
                               textposition='top center', name=f'{label}-axis', showlegend=False))
    return traces


def scene_layout(results, board_width_m, board_height_m):
    P = np.array([np.asarray(r['camera_position'], float) for r in results])
    half = max(np.ptp(P[:, 0]), np.ptp(P[:, 1]), np.ptp(P[:, 2]), board_width_m, board_height_m) * 0.6 + 0.2
    centre = ((P[:, 0].mean() + board_width_m / 2) / 2, (P[:, 1].mean() + board_height_m / 2) / 2, P[:, 2].mean() / 2)
    axis = lambda name, c: dict(title=dict(text=f'{name} (m)'), range=[float(c - half), float(c + half)])
    return dict(
        title=dict(text=TITLE, font=dict(size=20)),
        scene=dict(xaxis=axis('X', centre[0]), yaxis=axis('Y', centre[1]), zaxis=axis('Z', centre[2]), aspectmode='cube',
                   camera=dict(eye=dict(x=1.5, y=1.5, z=1.5))),
        legend=dict(yanchor='top', y=0.99, xanchor='left', x=0.01, bgcolor='rgba(255, 255, 255, 0.8)'),
        margin=dict(l=0, r=0, t=50, b=0), showlegend=True)


def _plain(o):
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    if isinstance(o, np.generic):
        return o.item()
    return o


def write_scene_html(results, board_width_m, board_height_m, output_html):
    """Writes the scene to `output_html` (folders created) and returns the figure specification
    ``dict(data=[...], layout={...})``; None when there is no pose to draw (main.py:236-238).  Never opens a browser."""
    if not results:
        print("❌ No valid poses to visualize")
        return None
    spec = _plain(dict(data=scene_traces(results, board_width_m, board_height_m),
                       layout=scene_layout(results, board_width_m, board_height_m)))
    os.makedirs(os.path.dirname(os.path.abspath(output_html)), exist_ok=True)
    try:
        import plotly.graph_objects as go
        go.Figure(spec).write_html(output_html)
    except ImportError:     # no Plotly: the same figure as a page that loads plotly.js itself
        with open(output_html, "w", encoding="utf-8") as f:
            f.write("<html><head><meta charset='utf-8'/><script src='https://cdn.plot.ly/plotly-2.35.2.min.js'></script></head><body>"
                    "<div id='scene' style='width:100%;height:100vh'></div><script>const spec = " + json.dumps(spec) +
                    "; Plotly.newPlot('scene', spec.data, spec.layout);</script></body></html>")
    print(f"\n✅ Interactive 3D visualization saved to: {output_html}")
    return spec
