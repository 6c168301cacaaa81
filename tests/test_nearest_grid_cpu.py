"""The guarantee behind the grid of index ranges (nearest_index_grid), checked on the CPU with a
NumPy restatement of the table construction (cl_grid_meta_kernel / cl_grid_cells_kernel): for every query
point inside the grid, the first minimum of the squared distance over the cell's index range [lo, hi] is the
first minimum over ALL candidate points 0 .. S-2 (car_dynamics.py:185-190: strict "<", in index order) -- on
the straight line, a closed circle, a zig-zag, an integer row with exact ties and the Bezier lane changes.
(The HIP code itself is checked on the GPU: tests/test_gpu_parity.py::test_block_pruned_nearest_point_is_exact.)"""
import numpy as np
import pytest

from model_predictive_control_amd import bezier_curves as bc

GRID_CELLS = 65536


def build_grid(x, y):
    """meta (x0, y0, cell, nx, ny) and the ranges lo, hi [ny, nx] of one centerline row (candidates 0 .. S-2)."""
    S = x.size
    nc = S - 1
    cx, cy = x[:nc], y[:nc]
    sp = np.hypot(np.diff(cx), np.diff(cy)).sum() / (nc - 1)
    R = 25.0 * sp
    W, H = (cx.max() - cx.min()) + 2 * R, (cy.max() - cy.min()) + 2 * R
    cell = 0.5 * sp
    nx, ny = np.ceil(W / cell), np.ceil(H / cell)
    for _ in range(64):
        if nx * ny <= GRID_CELLS:
            break
        cell *= 1.02 * np.sqrt(nx * ny / GRID_CELLS)
        nx, ny = np.ceil(W / cell), np.ceil(H / cell)
    nx, ny = int(nx), int(ny)
    x0, y0 = cx.min() - R, cy.min() - R
    e = 1e-6 * cell
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny))
    rx0, rx1 = x0 + ix * cell - e, x0 + (ix + 1) * cell + e
    ry0, ry1 = y0 + iy * cell - e, y0 + (iy + 1) * cell + e
    lo = np.full((ny, nx), nc, dtype=np.int64)
    hi = np.full((ny, nx), -1, dtype=np.int64)
    U = np.full((ny, nx), np.inf)
    for i in range(nc):                                    # U = min_j (largest squared distance from the cell to point j)
        ax = np.maximum(np.abs(cx[i] - rx0), np.abs(cx[i] - rx1))
        ay = np.maximum(np.abs(cy[i] - ry0), np.abs(cy[i] - ry1))
        U = np.minimum(U, ax * ax + ay * ay)
    bound = U * (1.0 + 1e-9)
    for i in range(nc):                                    # points whose smallest squared distance to the cell is <= U
        bx = np.maximum(np.maximum(rx0 - cx[i], cx[i] - rx1), 0.0)
        by = np.maximum(np.maximum(ry0 - cy[i], cy[i] - ry1), 0.0)
        ok = bx * bx + by * by <= bound
        lo = np.where(ok & (lo > i), i, lo)
        hi = np.where(ok, i, hi)
    return (x0, y0, cell, nx, ny), lo, hi


def full_scan(cx, cy, px, py):
    d = (cx[None, :] - px[:, None]) ** 2 + (cy[None, :] - py[:, None]) ** 2
    return np.argmin(d, axis=1)                            # first index of the minimum, as the strict "<" scan


def rows(S):
    th = np.linspace(0, 2 * np.pi, S)
    out = [(np.arange(S) / 10 - 0.1, np.zeros(S)),
           (5 * np.cos(th), 5 * np.sin(th) + 5),
           (np.arange(S, dtype=float), np.zeros(S)),
           (np.arange(S) * 0.25, (np.arange(S) % 3) * 0.5)]
    for r in bc.lane_change_centerlines(S=S)[::3]:
        out.append((r[:S].copy(), r[S:].copy()))
    return out


@pytest.mark.parametrize("S", [100, 37])
def test_cell_ranges_contain_the_scan_argmin(S):
    rng = np.random.default_rng(S)
    for x, y in rows(S):
        (x0, y0, cell, nx, ny), lo, hi = build_grid(x, y)
        assert nx * ny <= GRID_CELLS and (lo <= hi).all()
        cx, cy = x[:S - 1], y[:S - 1]
        n = 20000
        px = rng.uniform(x0, x0 + nx * cell, n)
        py = rng.uniform(y0, y0 + ny * cell, n)
        # exact ties and cell borders: midway between consecutive points, and points on cell corners
        k = min(S - 2, 60)
        px[:k], py[:k] = 0.5 * (cx[:k] + cx[1:k + 1]), 0.5 * (cy[:k] + cy[1:k + 1])
        px[k:2 * k] = x0 + cell * rng.integers(1, nx, k)
        py[k:2 * k] = y0 + cell * rng.integers(1, ny, k)
        fx, fy = (px - x0) * (1.0 / cell), (py - y0) * (1.0 / cell)
        inside = (fx >= 0) & (fx < nx) & (fy >= 0) & (fy < ny)
        ix, iy = fx[inside].astype(np.int64), fy[inside].astype(np.int64)
        want = full_scan(cx, cy, px[inside], py[inside])
        l, h = lo[iy, ix], hi[iy, ix]
        assert ((l <= want) & (want <= h)).all()
        # ... and the first minimum over the range IS that index
        d = (cx[None, :] - px[inside][:, None]) ** 2 + (cy[None, :] - py[inside][:, None]) ** 2
        idx = np.arange(S - 1)[None, :]
        d = np.where((idx >= l[:, None]) & (idx <= h[:, None]), d, np.inf)
        assert np.array_equal(np.argmin(d, axis=1), want)
        # the point of it: a lane looks at a handful of points, not at S - 1
        assert (h - l + 1).mean() < 0.5 * (S - 1)
