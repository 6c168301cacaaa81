// mpc_aux.hpp -- small standalone kernels behind the C-ABI: the model-layer entry points
// (a-1 .. a-6), the K2 / K3 test harnesses and the closed-loop plant step (f-1).
#pragma once
#include "mpc_eval.hpp"

namespace mpc {

// a-1 standalone: dx = f(x, u), agent-major arrays
template <int MODEL>
__global__ void __launch_bounds__(64) rhs_kernel(const DevCfg c, int B, const double *__restrict__ x,
                           const double *__restrict__ u, double *__restrict__ dx)
{
    constexpr int NX = ModelDim<MODEL>::NX;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= B) return;
    double xv[NX], k[NX];
    for (int i = 0; i < NX; i++) xv[i] = x[(size_t)a * NX + i];
    StageInput<MODEL> s;
    prep_input(c, u[2 * (size_t)a], u[2 * (size_t)a + 1], s);
    Lin<MODEL> dummy;
    rhs<false>(c, s, xv, k, dummy);
    for (int i = 0; i < NX; i++) dx[(size_t)a * NX + i] = k[i];
}

// a-2/a-3 standalone: X[B][Nsim][nx]
template <int MODEL>
__global__ void __launch_bounds__(64) simulate_kernel(const DevCfg c, int B, int Nsim, const double *__restrict__ x0,
                               const double *__restrict__ U, double *__restrict__ X)
{
    constexpr int NX = ModelDim<MODEL>::NX;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= B) return;
    double xv[NX];
    for (int i = 0; i < NX; i++) xv[i] = x0[(size_t)a * NX + i];
    for (int n = 0; n < Nsim; n++) {
        StageInput<MODEL> s;
        prep_input(c, U[((size_t)a * Nsim + n) * 2], U[((size_t)a * Nsim + n) * 2 + 1], s);
        stage_forward<MODEL>(c, s, xv);
        for (int i = 0; i < NX; i++) X[((size_t)a * Nsim + n) * NX + i] = xv[i];
    }
}

// bounding boxes of the blocks of NEAR_BLK consecutive candidate points of every centerline
// row (candidates are the points 0 .. S-2): boxes[row][block] = [xlo, xhi, ylo, yhi]
__global__ void cl_blocks_kernel(const DevCfg c, const double *__restrict__ cl, int C, double *__restrict__ boxes)
{
    const int NB = (c.S - 1 + NEAR_BLK - 1) / NEAR_BLK;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C * NB) return;
    const int row = t / NB, b = t - row * NB;
    const double *x = cl + (size_t)row * 2 * (size_t)c.S, *y = x + c.S;
    const int i0 = b * NEAR_BLK, i1 = min(i0 + NEAR_BLK, c.S - 1);
    double xlo = x[i0], xhi = x[i0], ylo = y[i0], yhi = y[i0];
    for (int i = i0 + 1; i < i1; i++) {
        xlo = fmin(xlo, x[i]); xhi = fmax(xhi, x[i]); ylo = fmin(ylo, y[i]); yhi = fmax(yhi, y[i]);
    }
    double *q = boxes + 4 * (size_t)t;
    q[0] = xlo; q[1] = xhi; q[2] = ylo; q[3] = yhi;
}

// placement of the nearest-point grid of every centerline row (one wave per row): the box of the
// candidate points grown by 25 mean spacings, cells of half a spacing (larger if that takes more than
// GRID_CELLS cells).  A row the grid cannot describe (non-finite or coincident points, S - 1 > 65535)
// gets nx = ny = 0: every query is "outside" and takes the full scan.  Also writes the interleaved copy.
__global__ void __launch_bounds__(64) cl_grid_meta_kernel(const DevCfg c, const double *__restrict__ cl, int C,
                                                          double *__restrict__ meta, double *__restrict__ xy)
{
    const int row = blockIdx.x, lane = threadIdx.x;
    if (row >= C) return;
    const int S = c.S, nc = S - 1;
    const double *x = cl + (size_t)row * 2 * (size_t)S, *y = x + S;
    double *q = xy + (size_t)row * 2 * (size_t)S;
    for (int i = lane; i < S; i += 64) { q[2 * i] = x[i]; q[2 * i + 1] = y[i]; }
    double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY, len = 0.0;
    bool ok = true;
    for (int i = lane; i < nc; i += 64) {
        ok = ok && isfinite(x[i]) && isfinite(y[i]);
        xlo = fmin(xlo, x[i]); xhi = fmax(xhi, x[i]); ylo = fmin(ylo, y[i]); yhi = fmax(yhi, y[i]);
        if (i) len += sqrt((x[i] - x[i - 1]) * (x[i] - x[i - 1]) + (y[i] - y[i - 1]) * (y[i] - y[i - 1]));
    }
    for (int o = 32; o; o >>= 1) {
        xlo = fmin(xlo, __shfl_xor(xlo, o)); xhi = fmax(xhi, __shfl_xor(xhi, o));
        ylo = fmin(ylo, __shfl_xor(ylo, o)); yhi = fmax(yhi, __shfl_xor(yhi, o));
        len += __shfl_xor(len, o);
    }
    ok = __ballot(!ok) == 0ull && nc >= 1 && nc <= 65535;
    if (lane) return;
    const double sp = nc > 1 ? len / (double)(nc - 1) : 1.0;
    ok = ok && isfinite(sp) && sp > 0.0;
    double *m = meta + (size_t)row * GRID_META;
    for (int i = 0; i < GRID_META; i++) m[i] = 0.0;
    if (!ok) return;
    const double R = 25.0 * sp, W = (xhi - xlo) + 2.0 * R, H = (yhi - ylo) + 2.0 * R;
    double cell = 0.5 * sp, nx = ceil(W / cell), ny = ceil(H / cell);
    for (int it = 0; it < 64 && nx * ny > (double)GRID_CELLS; it++) {
        cell *= 1.02 * sqrt(nx * ny / (double)GRID_CELLS);
        nx = ceil(W / cell); ny = ceil(H / cell);
    }
    if (!(nx * ny <= (double)GRID_CELLS) || !isfinite(1.0 / cell)) return;
    m[0] = xlo - R; m[1] = ylo - R; m[2] = 1.0 / cell; m[3] = nx; m[4] = ny; m[5] = cell;
}

// index range of one grid cell (one thread per cell): see nearest_index_grid
__global__ void __launch_bounds__(256) cl_grid_cells_kernel(const DevCfg c, const double *__restrict__ cl, int C,
                                                            const double *__restrict__ meta, unsigned *__restrict__ cells)
{
    const int row = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    const double *m = meta + (size_t)row * GRID_META;
    const int nx = (int)m[3], ny = (int)m[4];
    if (t >= nx * ny) return;
    const int S = c.S, nc = S - 1;
    const double *x = cl + (size_t)row * 2 * (size_t)S, *y = x + S;
    const double cell = m[5], e = 1e-6 * cell;
    const int ix = t % nx, iy = t / nx;
    const double rx0 = m[0] + ix * cell - e, rx1 = m[0] + (ix + 1) * cell + e;
    const double ry0 = m[1] + iy * cell - e, ry1 = m[1] + (iy + 1) * cell + e;
    double U = INFINITY;
    for (int i = 0; i < nc; i++) {
        const double ax = fmax(fabs(x[i] - rx0), fabs(x[i] - rx1)), ay = fmax(fabs(y[i] - ry0), fabs(y[i] - ry1));
        U = fmin(U, ax * ax + ay * ay);
    }
    const double bound = U * (1.0 + 1e-9);
    int lo = nc, hi = -1;
    for (int i = 0; i < nc; i++) {
        const double bx = fmax(fmax(rx0 - x[i], x[i] - rx1), 0.0), by = fmax(fmax(ry0 - y[i], y[i] - ry1), 0.0);
        if (bx * bx + by * by <= bound) { lo = min(lo, i); hi = i; }
    }
    if (hi < 0) { lo = 0; hi = nc - 1; }      // (cannot happen: the point that attains U passes)
    cells[(size_t)row * GRID_CELLS + t] = (unsigned)lo | ((unsigned)hi << 16);
}

// a-4/a-5 standalone (nt: the tables of the pruned searches, all null for the full scan)
__global__ void errors_kernel(const DevCfg c, int B, const double *__restrict__ pose,
                              const double *__restrict__ cl, const int *__restrict__ cl_index,
                              const NearTab nt, double *__restrict__ err, int *__restrict__ idx_out)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    // (no early return: the block search's loop is wave-uniform; lanes past the batch redo agent 0)
    const bool live = a < B;
    const int aa = live ? a : 0;
    const int row = cl_index ? cl_index[aa] : 0;
    const double *clp = cl + (size_t)row * 2 * (size_t)c.S;
    const double px = pose[(size_t)aa * 3], py = pose[(size_t)aa * 3 + 1], phi = pose[(size_t)aa * 3 + 2];
    const int idx = nearest_lookup(c, clp, nt, row, px, py);
    if (!live) return;
    Geom g;
    load_geom(c, clp, idx, g);
    double cte, he, pe;
    tracking_errors(c, g, px, py, phi, cte, he, pe);
    err[(size_t)a * 3] = cte; err[(size_t)a * 3 + 1] = he; err[(size_t)a * 3 + 2] = pe;
    if (idx_out) idx_out[a] = idx;
}

// a-6 standalone: L[b] = stage cost of (x[b], u[b]) against its centerline (car_dynamics.py:252-258)
template <int MODEL>
__global__ void __launch_bounds__(64) stage_cost_kernel(const DevCfg c, int B, const double *__restrict__ x,
                                  const double *__restrict__ u, const double *__restrict__ cl,
                                  const int *__restrict__ cl_index, double *__restrict__ out)
{
    constexpr int NX = ModelDim<MODEL>::NX;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= B) return;
    const double *clp = cl + (size_t)(cl_index ? cl_index[a] : 0) * 2 * (size_t)c.S;
    double xv[NX], xb[NX], ub[2];
    for (int i = 0; i < NX; i++) xv[i] = x[(size_t)a * NX + i];
    const int idx = nearest_index(c, clp, xv[0], xv[1]);
    Geom g;
    load_geom(c, clp, idx, g);
    out[a] = stage_cost<MODEL, false>(c, g, xv, u[2 * (size_t)a], u[2 * (size_t)a + 1], xb, ub);
}

// K2 standalone (agent-major arrays)
__global__ void prox_kernel(const DevCfg c, int B, const double *__restrict__ x,
                            const double *__restrict__ g, const double *__restrict__ gamma,
                            double *__restrict__ xhat, double *__restrict__ p,
                            double *__restrict__ out)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= B) return;
    double pp = 0.0, gp = 0.0;
    const double gm = gamma[a];
    for (int j = 0; j < c.n; j++) {
        const double xv = x[(size_t)a * c.n + j], gv = g[(size_t)a * c.n + j];
        const double pv = prox_p(c, j & 1, xv, gv, gm);
        if (xhat) xhat[(size_t)a * c.n + j] = xv + pv;
        if (p) p[(size_t)a * c.n + j] = pv;
        pp += pv * pv; gp += gv * pv;
    }
    out[2 * (size_t)a] = pp; out[2 * (size_t)a + 1] = gp;
}

// closed loop helpers (main.py:141-146): u0 = U[:, 0], x <- f_d(x, u0), optional warm-start shift
template <int MODEL>
__global__ void __launch_bounds__(64) plant_step_kernel(const DevCfg c, int B, int t, int T, int shift,
                                  double *__restrict__ x, double *__restrict__ U,
                                  double *__restrict__ traj_x, double *__restrict__ traj_u,
                                  const double *__restrict__ stats, int *__restrict__ fail_count)
{
    constexpr int NX = ModelDim<MODEL>::NX;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= B) return;
    double xv[NX];
    for (int i = 0; i < NX; i++) xv[i] = x[(size_t)a * NX + i];
    double *Ua = U + (size_t)a * c.n;
    const double d = Ua[0], dl = Ua[1];
    StageInput<MODEL> s;
    prep_input(c, d, dl, s);
    stage_forward<MODEL>(c, s, xv);
    for (int i = 0; i < NX; i++) {
        x[(size_t)a * NX + i] = xv[i];
        if (traj_x) traj_x[((size_t)a * T + t) * NX + i] = xv[i];
    }
    if (traj_u) { traj_u[((size_t)a * T + t) * 2] = d; traj_u[((size_t)a * T + t) * 2 + 1] = dl; }
    if (shift) {
        for (int j = 0; j + 2 < c.n; j++) Ua[j] = Ua[j + 2];
    }
    if (fail_count && stats) fail_count[a] += stats[(size_t)a * 8] != (double)ST_CONVERGED;
}


// device-math probe (test only): op 0 sin, 1 cos, 2 atan, 3 atan2(a, b), 4 tan
__global__ void math_probe_kernel(int n, int op, const double *__restrict__ a, const double *__restrict__ b,
                                  double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (op) {
    case 0: r = m_sincos(a[i]).s; break;
    case 1: r = m_sincos(a[i]).c; break;
    case 2: r = m_atan(a[i]); break;
    case 3: r = m_atan2(a[i], b[i]); break;
    default: r = m_tan(a[i]); break;
    }
    out[i] = r;
}

// K3 harness: one wave per agent on agent-major S, Y [B][M][n], mask, q [B][n]
template <int NE, int MC>
__global__ void __launch_bounds__(256)
lbfgs_apply_kernel(const DevCfg c, int B, const double *__restrict__ S, const double *__restrict__ Y,
                   const int *__restrict__ idx, const int *__restrict__ full,
                   const double *__restrict__ mask, double *__restrict__ q, int *__restrict__ ok,
                   unsigned long long *rows)
{
    (void)rows;
    int rows_read = 0;
    const int lane = threadIdx.x & 63;
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= B) return;
    const int n = c.n;
    Row<NE> qv = ldrow<NE>(q + (size_t)a * n, n, lane);
    const Row<NE> mk = ldrow<NE>(mask + (size_t)a * n, n, lane);
    bool inj[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) inj[e] = lane + 64 * e < n && mk.v[e] != 0.0;
    const bool r = lbfgs_two_loop<NE, MC>(c, S + (size_t)a * c.M * n, Y + (size_t)a * c.M * n, n, lane, inj,
                                          idx[a], full[a], qv, rows_read);
    if (r) strow<NE>(q + (size_t)a * n, n, lane, qv);
    if (lane == 0) ok[a] = r ? 1 : 0;
}

} // namespace mpc
