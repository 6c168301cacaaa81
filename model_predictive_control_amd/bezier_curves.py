"""Lane-change centerline generator for BASELINE.json's config 3 (per-agent lane-change
centerlines).

Same curves as the reference's bezier_curves.py:19-48 (quintic Bernstein curve whose control
points come from the overtaking geometry), stated in closed form: every control polygon of the
family is `Px = Px2 * [0, 1/i, 1, 1, 2 - 1/i, 2]`, `Py = h * [0, 0, 0, 1, 1, 1]`, so the table for
all shapes `i` is one broadcast, and a curve is one (samples x 6) Bernstein basis times the
polygon.  `get_bezier_control_points(i)` / `bezier_curve(j, P)` keep the reference's call
signatures on top of that.  Host-side NumPy (input generation runs once per scenario, as in the
reference); the HIP kernels consume the rows [x_0..x_{S-1}, y_0..y_{S-1}] selected per agent by
`cl_index`.
"""
from math import comb

import numpy as np

# overtaking geometry, values of bezier_curves.py:5-12 (module-level names kept for callers)
h = 3.75
L, W = 4.2, 1.8
θ = 3.2 / 180 * np.pi
l = 3
Lf = 1
v0, v1 = 20, 10
D1 = 50

_DEGREE = 5
_BINOM = np.array([comb(_DEGREE, k) for k in range(_DEGREE + 1)], dtype=np.float64)
_K = np.arange(_DEGREE + 1)


def binomial_coefficient(n, k):
    """bezier_curves.py:15-16 (a float, like the reference's division of factorials)."""
    return float(comb(n, k))


def overtake_abscissa():
    """(Px2, tca): where the ego car reaches the lead car's rear, and the time that takes
    (bezier_curves.py:32-35,:47)."""
    closing = v0 - v1
    clearance = (Lf + l) * np.cos(np.arctan2(W, 2 * Lf) - θ)
    px2 = v0 * (D1 / closing) - clearance
    return px2, px2 / closing


def control_point_table(shapes):
    """Control polygons [len(shapes), 2, 6] of the curve family and the common lane-change time."""
    px2, tca = overtake_abscissa()
    step = px2 / np.asarray(list(shapes), dtype=np.float64)     # Px2 / i, one per shape
    one = np.ones_like(step)
    # Px1 = Px2 / i, Px4 = 2 Px2 - Px2 / i  (the reference's (Px5 - Px3) is exactly Px2)
    px = np.stack([0 * one, step, px2 * one, px2 * one, 2 * px2 - step, 2 * px2 * one], 1)
    py = np.broadcast_to(h * (_K >= 3), px.shape)
    return np.stack([px, py], 1), tca


def bernstein_basis(j):
    """B[..., k] = C(5, k) (1 - j)^(5 - k) j^k for k = 0..5."""
    j = np.asarray(j, dtype=np.float64)[..., None]
    return _BINOM * (1 - j) ** (_DEGREE - _K) * j ** _K


def bezier_curve(j, P):
    """Point(s) of the quintic curve with control polygon P (2, 6) at parameter j (scalar or array)."""
    B = bernstein_basis(j)
    P = np.asarray(P, dtype=np.float64)
    return (B * P[0]).sum(-1), (B * P[1]).sum(-1)


def get_bezier_control_points(i):
    """Control points (2, 6) of shape i and the lane-change time tca (bezier_curves.py:28-48)."""
    table, tca = control_point_table([i])
    return table[0], tca


def lane_change_centerlines(S=100, scale=10.0 / 193.76417765201978, shapes=range(1, 11)):
    """Table [len(shapes), 2S] of lane-change centerlines, one per curve shape i (the reference plots
    i = 1..10, bezier_curves.py:56-59), sampled at S uniform parameters and scaled from the
    reference's highway dimensions (194 m x 3.75 m) to the 1:43 car of main.py:82-86 (build-defined:
    the default maps the curve onto 10 m x 0.19 m)."""
    table, _ = control_point_table(shapes)
    B = bernstein_basis(np.linspace(0.0, 1.0, S))               # [S, 6]
    xy = np.einsum("sk,cdk->cds", B, table) * scale             # [C, 2, S]
    return xy.reshape(xy.shape[0], 2 * S)
