/*
 * mpc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's batched MPC solve step.  Every
 * function cites the reference file:line it follows (paths are into
 * /root/reference).  The solver arithmetic lives in alpaqa (pip dependency,
 * not vendored, version unpinned; API fingerprint = 0.0.1): it is restated
 * from the published algorithm (Pas, Schuurmans, Patrinos, "Alpaqa", ECC 2022)
 * and anchored on the reference call sites controller.py:27-48,57 and
 * main.py:54-56.  PARITY STATUS of the solver layer: "parity unpinned".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 */
#include "mpc_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXNX 6
#define PI_D 3.14159265358979323846

/* ------------------------------------------------------------------ config */

void orc_default_config(orc_config *c, int model, int N)
{
    memset(c, 0, sizeof(*c));
    c->model = model;
    c->N = N;
    c->S = 100;                 /* main.py:70 */
    c->nfe = 4;                 /* car_dynamics.py:136 */
    c->wrap_mode = ORC_WRAP_FLOOR;
    c->clip_inputs = 0;
    c->constr_mode = ORC_CONSTR_NONE;
    c->lbfgs_memory = N;        /* controller.py:36 */
    c->max_iter = 1000;         /* controller.py:31 */
    c->max_outer = 1000;        /* controller.py:45 */
    c->hess_heuristic = 15;     /* controller.py:32 */
    c->max_no_progress = 10;
    c->Ts = 0.05;               /* car_dynamics.py:93 */
    c->v_ref = 1.0;             /* main.py:65 */
    const double w[6] = {0.5, 1.0, 1.0, 0.5, 0.1, 0.01}; /* car_dynamics.py:230 */
    memcpy(c->cost_w, w, sizeof w);
    /* main.py:82-111, order car_dynamics.py:65-88 */
    const double veh[22] = {9.7e-2, 4.7e-2, 5e-2, 0.09, 0.07, 8e-2, 5.5e-2, 0.1735, 18.3e-5,
                            0.32, 1.0, 0.268, 2.165, 3.47, 0.242, 2.38, 2.84,
                            0.266, 0.1, 0.1025, 0.1629, 0.0011};
    memcpy(c->veh, veh, sizeof veh);
    c->accel = 2.0;             /* dynamics.py:35 */
    c->friction = 1.0;          /* dynamics.py:34 */
    c->u_lb[0] = -1.0; c->u_ub[0] = 1.0;    /* main.py:55-56, :82 */
    c->u_lb[1] = -0.32; c->u_ub[1] = 0.32;
    const double off[6] = {20, 1, 1, 2, 1, 0.1}; /* main.py:46-51 */
    memcpy(c->g_off, off, sizeof off);
    for (int i = 0; i < 6; i++) { c->D_lb[i] = -INFINITY; c->D_ub[i] = INFINITY; } /* main.py:57 */
    c->lane_halfwidth = 0.15;
    c->alm_eps = 1e-6; c->alm_delta = 1e-4; c->Sigma0 = 1e5;   /* controller.py:41-43 */
    c->eps0 = 1.0; c->rho = 0.1; c->Delta = 10.0; c->theta = 0.1;
    c->M = 1e9; c->Sigma_max = 1e9;
    c->Delta_lower = 0.8; c->Sigma0_lower = 0.6; c->eps0_increase = 1.1; c->rho_increase = 2.0;
    c->max_num_initial_retries = 20; c->max_num_retries = 20; c->max_total_num_retries = 40;
    c->max_total_inner = 5000; c->max_total_evals = 0;
    c->lip_eps = 1e-6; c->lip_delta = 1e-12; c->Lgamma_factor = 0.95;
    c->L_min = 1e-5; c->L_max = 1e20; c->tau_min = 1.0 / 256; c->qub_tol = 10 * DBL_EPSILON;
}

int orc_nx(const orc_config *c) { return c->model == ORC_MODEL_PACEJKA ? 6 : 4; }

static int stage_m(const orc_config *c)
{
    if (c->constr_mode == ORC_CONSTR_STATE_SQ) return orc_nx(c);
    if (c->constr_mode == ORC_CONSTR_LANE) return 1;
    return 0;
}
int orc_m(const orc_config *c) { return stage_m(c) * c->N; }

/* ------------------------------------------------------------------ model */

static void clip_u(const orc_config *c, const double *u, double *ue, double *mask)
{
    /* dynamics.py:57-65 clip_inputs: d by max_drive (veh[10]), delta by max_steer (veh[9]) */
    double lim[2] = {c->veh[10], c->veh[9]};
    for (int i = 0; i < 2; i++) {
        ue[i] = u[i]; mask[i] = 1.0;
        if (c->clip_inputs) {
            if (u[i] > lim[i]) { ue[i] = lim[i]; mask[i] = 0.0; }
            else if (u[i] < -lim[i]) { ue[i] = -lim[i]; mask[i] = 0.0; }
        }
    }
}

static double sgn(double v) { return (v > 0) - (v < 0); }

/* a-1: car_dynamics.py:93-129 (== dynamics.py:67-119); a-1b: dynamics.py:144-173 */
void orc_rhs(const orc_config *c, const double *x, const double *u_in, double *dx)
{
    double u[2], mk[2];
    clip_u(c, u_in, u, mk);
    const double lf = c->veh[1], lr = c->veh[2];
    if (c->model == ORC_MODEL_PACEJKA) {
        const double m = c->veh[7], iz = c->veh[8];
        const double bf = c->veh[11], cf = c->veh[12], df = c->veh[13];
        const double br = c->veh[14], cr = c->veh[15], dr = c->veh[16];
        const double cm1 = c->veh[17], cm2 = c->veh[18], cr0 = c->veh[19], cr2 = c->veh[21];
        const double d = u[0], dl = u[1];
        const double phi = x[2], vx = x[3], vy = x[4], om = x[5];
        double af = -atan2(om * lf + vy, vx) + dl;
        double ar = atan2(om * lr - vy, vx);
        double frx = (cm1 - cm2 * vx) * d - cr0 * sgn(vx) - cr2 * vx * vx;
        double ffy = df * sin(cf * atan(bf * af));
        double fry = dr * sin(cr * atan(br * ar));
        dx[0] = vx * cos(phi) - vy * sin(phi);
        dx[1] = vx * sin(phi) + vy * cos(phi);
        dx[2] = om;
        dx[3] = (frx - ffy * sin(dl) + m * vy * om) / m;
        dx[4] = (fry + ffy * cos(dl) - m * vx * om) / m;
        dx[5] = (ffy * lf * cos(dl) - fry * lr) / iz;
    } else {
        const double phi = x[2], v = x[3];
        double beta = atan2(lf * tan(u[1]), lf + lr);
        dx[0] = v * cos(phi + beta);
        dx[1] = v * sin(phi + beta);
        dx[2] = v * sin(beta) / lr;
        dx[3] = c->accel * u[0] - c->friction * v;
    }
}

/* vector-Jacobian product of orc_rhs: yb += J_x^T w, ub += J_u^T w
 * (sign(vx) is a constant under AD, SURVEY 7 "non-smooth ops") */
static void rhs_vjp(const orc_config *c, const double *x, const double *u_in, const double *w,
                    double *yb, double *ub)
{
    double u[2], mk[2];
    clip_u(c, u_in, u, mk);
    const double lf = c->veh[1], lr = c->veh[2];
    if (c->model == ORC_MODEL_PACEJKA) {
        const double m = c->veh[7], iz = c->veh[8];
        const double bf = c->veh[11], cf = c->veh[12], df = c->veh[13];
        const double br = c->veh[14], cr = c->veh[15], dr = c->veh[16];
        const double cm1 = c->veh[17], cm2 = c->veh[18], cr2 = c->veh[21];
        const double d = u[0], dl = u[1];
        const double phi = x[2], vx = x[3], vy = x[4], om = x[5];
        const double sp = sin(phi), cp = cos(phi), sd = sin(dl), cd = cos(dl);
        const double a1 = om * lf + vy, a2 = om * lr - vy;
        const double af = dl - atan2(a1, vx), ar = atan2(a2, vx);
        const double tf = atan(bf * af), tr = atan(br * ar);
        const double ffy = df * sin(cf * tf);
        /* adjoints of the three forces */
        double frx_b = w[3] / m;
        double ffy_b = -w[3] * sd / m + w[4] * cd / m + w[5] * lf * cd / iz;
        double fry_b = w[4] / m - w[5] * lr / iz;
        double phi_b = w[0] * (-vx * sp - vy * cp) + w[1] * (vx * cp - vy * sp);
        double vx_b = w[0] * cp + w[1] * sp - w[4] * om;
        double vy_b = -w[0] * sp + w[1] * cp + w[3] * om;
        double om_b = w[2] + w[3] * vy - w[4] * vx;
        double dl_b = -w[3] * ffy * cd / m - w[4] * ffy * sd / m - w[5] * ffy * lf * sd / iz;
        double d_b = frx_b * (cm1 - cm2 * vx);
        vx_b += frx_b * (-cm2 * d - 2.0 * cr2 * vx);
        double af_b = ffy_b * df * cos(cf * tf) * cf * bf / (1.0 + bf * af * bf * af);
        double ar_b = fry_b * dr * cos(cr * tr) * cr * br / (1.0 + br * ar * br * ar);
        double r1 = a1 * a1 + vx * vx, r2 = a2 * a2 + vx * vx;
        dl_b += af_b;
        double a1_b = -af_b * vx / r1;
        vx_b += af_b * a1 / r1;
        om_b += a1_b * lf; vy_b += a1_b;
        double a2_b = ar_b * vx / r2;
        vx_b += -ar_b * a2 / r2;
        om_b += a2_b * lr; vy_b -= a2_b;
        yb[2] += phi_b; yb[3] += vx_b; yb[4] += vy_b; yb[5] += om_b;
        ub[0] += d_b * mk[0]; ub[1] += dl_b * mk[1];
    } else {
        const double phi = x[2], v = x[3];
        const double L = lf + lr;
        const double td = tan(u[1]);
        const double t = lf * td;
        const double beta = atan2(t, L);
        const double s = sin(phi + beta), co = cos(phi + beta);
        double phi_b = w[0] * (-v * s) + w[1] * (v * co);
        double v_b = w[0] * co + w[1] * s + w[2] * sin(beta) / lr - w[3] * c->friction;
        double beta_b = w[0] * (-v * s) + w[1] * (v * co) + w[2] * v * cos(beta) / lr;
        double dl_b = beta_b * (L / (t * t + L * L)) * lf * (1.0 + td * td);
        double d_b = w[3] * c->accel;
        yb[2] += phi_b; yb[3] += v_b;
        ub[0] += d_b * mk[0]; ub[1] += dl_b * mk[1];
    }
}

/* one classical RK4 step of size h, input held constant
 * (car_dynamics.py:136-145: cs.integrator("rk"), SURVEY a-2) */
static void rk4_step(const orc_config *c, const double *x, const double *u, double h, double *xn)
{
    const int nx = orc_nx(c);
    double k1[ORC_MAXNX], k2[ORC_MAXNX], k3[ORC_MAXNX], k4[ORC_MAXNX], t[ORC_MAXNX];
    orc_rhs(c, x, u, k1);
    for (int i = 0; i < nx; i++) t[i] = x[i] + 0.5 * h * k1[i];
    orc_rhs(c, t, u, k2);
    for (int i = 0; i < nx; i++) t[i] = x[i] + 0.5 * h * k2[i];
    orc_rhs(c, t, u, k3);
    for (int i = 0; i < nx; i++) t[i] = x[i] + h * k3[i];
    orc_rhs(c, t, u, k4);
    for (int i = 0; i < nx; i++) xn[i] = x[i] + (h / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
}

/* reverse of rk4_step: lam (in: adjoint of xn, out: adjoint of x), ub += adjoint of u */
static void rk4_step_vjp(const orc_config *c, const double *x, const double *u, double h,
                         double *lam, double *ub)
{
    const int nx = orc_nx(c);
    double k1[ORC_MAXNX], k2[ORC_MAXNX], k3[ORC_MAXNX];
    double y2[ORC_MAXNX], y3[ORC_MAXNX], y4[ORC_MAXNX];
    orc_rhs(c, x, u, k1);
    for (int i = 0; i < nx; i++) y2[i] = x[i] + 0.5 * h * k1[i];
    orc_rhs(c, y2, u, k2);
    for (int i = 0; i < nx; i++) y3[i] = x[i] + 0.5 * h * k2[i];
    orc_rhs(c, y3, u, k3);
    for (int i = 0; i < nx; i++) y4[i] = x[i] + h * k3[i];
    double kb[ORC_MAXNX], yb4[ORC_MAXNX] = {0}, yb3[ORC_MAXNX] = {0}, yb2[ORC_MAXNX] = {0},
           yb1[ORC_MAXNX] = {0};
    for (int i = 0; i < nx; i++) kb[i] = (h / 6.0) * lam[i];
    rhs_vjp(c, y4, u, kb, yb4, ub);
    for (int i = 0; i < nx; i++) kb[i] = (h / 3.0) * lam[i] + h * yb4[i];
    rhs_vjp(c, y3, u, kb, yb3, ub);
    for (int i = 0; i < nx; i++) kb[i] = (h / 3.0) * lam[i] + 0.5 * h * yb3[i];
    rhs_vjp(c, y2, u, kb, yb2, ub);
    for (int i = 0; i < nx; i++) kb[i] = (h / 6.0) * lam[i] + 0.5 * h * yb2[i];
    rhs_vjp(c, x, u, kb, yb1, ub);
    for (int i = 0; i < nx; i++) lam[i] += yb1[i] + yb2[i] + yb3[i] + yb4[i];
}

/* a-2: f_d = nfe RK4 steps of h = Ts/nfe (car_dynamics.py:136-145) */
void orc_fd(const orc_config *c, const double *x, const double *u, double *xn)
{
    const int nx = orc_nx(c);
    const double h = c->Ts / c->nfe;
    double a[ORC_MAXNX], b[ORC_MAXNX];
    memcpy(a, x, nx * sizeof(double));
    for (int s = 0; s < c->nfe; s++) { rk4_step(c, a, u, h, b); memcpy(a, b, nx * sizeof(double)); }
    memcpy(xn, a, nx * sizeof(double));
}

static void fd_vjp(const orc_config *c, const double *x, const double *u, double *lam, double *ub)
{
    const int nx = orc_nx(c);
    const double h = c->Ts / c->nfe;
    double sub[16][ORC_MAXNX]; /* nfe <= 16 */
    memcpy(sub[0], x, nx * sizeof(double));
    for (int s = 0; s + 1 < c->nfe; s++) rk4_step(c, sub[s], u, h, sub[s + 1]);
    for (int s = c->nfe - 1; s >= 0; s--) rk4_step_vjp(c, sub[s], u, h, lam, ub);
}

/* a-3: car_dynamics.py:159-166 simulate/mapaccum: columns x_1..x_N, U flat [d0,dl0,d1,dl1,..]
 * (car_dynamics.py:149-157 order='F') */
void orc_rollout(const orc_config *c, const double *x0, const double *U, double *X)
{
    const int nx = orc_nx(c);
    const double *cur = x0;
    for (int n = 0; n < c->N; n++) {
        orc_fd(c, cur, U + 2 * n, X + (size_t)n * nx);
        cur = X + (size_t)n * nx;
    }
}

/* a-4: car_dynamics.py:174-192.  cl is flat [x_0..x_{S-1}, y_0..y_{S-1}] (main.py:113).
 * Start at i=0, candidates i=1..S-2, strict <.  Squared distances are compared
 * (sqrt is monotone; see DESIGN.md for the 1-ulp tie caveat). */
int orc_nearest(const orc_config *c, const double *pos, const double *cl)
{
    const int S = c->S;
    double dx = cl[0] - pos[0], dy = cl[S] - pos[1];
    double best = dx * dx + dy * dy;
    int idx = 0;
    for (int i = 1; i < S - 1; i++) {
        dx = cl[i] - pos[0]; dy = cl[S + i] - pos[1];
        double d2 = dx * dx + dy * dy;
        if (d2 < best) { best = d2; idx = i; }
    }
    return idx;
}

typedef struct { double nx_, ny_, px_, py_, qx_, qy_; } geom3; /* nearest, previous, next */

static void geom_at(const orc_config *c, const double *pos, const double *cl, geom3 *g)
{
    const int S = c->S;
    int i = orc_nearest(c, pos, cl);
    int ip = i > 0 ? i - 1 : 0; /* car_dynamics.py:183: previous == nearest when index 0 wins */
    g->nx_ = cl[i]; g->ny_ = cl[S + i];
    g->px_ = cl[ip]; g->py_ = cl[S + ip];
    g->qx_ = cl[i + 1]; g->qy_ = cl[S + i + 1];
}

/* car_dynamics.py:168-172 */
static double wrap_to_pi(const orc_config *c, double ang)
{
    const double two_pi = 2.0 * PI_D;
    double a = ang + PI_D;
    double m;
    if (c->wrap_mode == ORC_WRAP_FMOD) m = fmod(a, two_pi);
    else if (c->wrap_mode == ORC_WRAP_IEEE) m = remainder(a, two_pi);
    else { m = fmod(a, two_pi); if (m < 0) m += two_pi; } /* numpy remainder, divisor > 0 */
    return m - PI_D;
}

static void errors_from_geom(const orc_config *c, const geom3 *g, const double *pos, double phi,
                             double *cte, double *he, double *pe)
{
    /* car_dynamics.py:212-214 */
    *cte = (pos[0] - g->px_) * (g->ny_ - g->py_) - (pos[1] - g->py_) * (g->nx_ - g->px_);
    /* car_dynamics.py:217-222: the is_equal test is structural -> always atan2 branch */
    double desired = atan2(g->qy_ - g->ny_, g->qx_ - g->nx_);
    *he = wrap_to_pi(c, desired - phi);
    /* car_dynamics.py:225-227 */
    *pe = (pos[0] - g->nx_) * (g->qy_ - g->ny_) - (pos[1] - g->ny_) * (g->qx_ - g->nx_);
}

/* a-5: car_dynamics.py:194-228 */
void orc_errors(const orc_config *c, const double *pos, double phi, const double *cl, double *out)
{
    geom3 g;
    geom_at(c, pos, cl, &g);
    errors_from_geom(c, &g, pos, phi, &out[0], &out[1], &out[2]);
}

static double speed_of(const orc_config *c, const double *x)
{
    /* car_dynamics.py:252; the nx=4 model has a single speed state (build-defined) */
    return c->model == ORC_MODEL_PACEJKA ? sqrt(x[3] * x[3] + x[4] * x[4]) : x[3];
}

/* a-6: car_dynamics.py:230-258.  If xb != NULL accumulate dL/dx into xb, dL/du into ub. */
static double stage_cost_g(const orc_config *c, const geom3 *g, const double *x, const double *u,
                           double *xb, double *ub)
{
    const double *w = c->cost_w;
    double cte, he, pe;
    errors_from_geom(c, g, x, x[2], &cte, &he, &pe);
    double sp = speed_of(c, x);
    double ev = sp - c->v_ref;
    double L = w[0] * ev * ev + w[1] * cte * cte + w[2] * pe * pe + w[3] * he * he +
               w[4] * u[1] * u[1] + w[5] * u[0] * u[0];
    if (xb) {
        xb[0] += 2.0 * w[1] * cte * (g->ny_ - g->py_) + 2.0 * w[2] * pe * (g->qy_ - g->ny_);
        xb[1] += -2.0 * w[1] * cte * (g->nx_ - g->px_) - 2.0 * w[2] * pe * (g->qx_ - g->nx_);
        xb[2] += -2.0 * w[3] * he;
        if (c->model == ORC_MODEL_PACEJKA) {
            xb[3] += 2.0 * w[0] * ev * x[3] / sp;
            xb[4] += 2.0 * w[0] * ev * x[4] / sp;
        } else {
            xb[3] += 2.0 * w[0] * ev;
        }
        ub[0] += 2.0 * w[5] * u[0];
        ub[1] += 2.0 * w[4] * u[1];
    }
    return L;
}

double orc_stage_cost(const orc_config *c, const double *x, const double *u, const double *cl)
{
    geom3 g;
    geom_at(c, x, cl, &g);
    return stage_cost_g(c, &g, x, u, NULL, NULL);
}

/* a-7 constraints of one stage: main.py:43-52 (STATE_SQ) or signed lateral distance (LANE) */
static void stage_constr(const orc_config *c, const geom3 *g, const double *x, double *gv)
{
    if (c->constr_mode == ORC_CONSTR_STATE_SQ) {
        int nx = orc_nx(c);
        for (int i = 0; i < nx; i++) gv[i] = x[i] * x[i] - c->g_off[i];
    } else if (c->constr_mode == ORC_CONSTR_LANE) {
        /* road.py:77-79: pos_error normalised by the segment length */
        double wx = g->qx_ - g->nx_, wy = g->qy_ - g->ny_;
        double pe = (x[0] - g->nx_) * wy - (x[1] - g->ny_) * wx;
        gv[0] = pe / sqrt(wx * wx + wy * wy);
    }
}

static void stage_constr_vjp(const orc_config *c, const geom3 *g, const double *x,
                             const double *yh, double *xb)
{
    if (c->constr_mode == ORC_CONSTR_STATE_SQ) {
        int nx = orc_nx(c);
        for (int i = 0; i < nx; i++) xb[i] += yh[i] * 2.0 * x[i];
    } else if (c->constr_mode == ORC_CONSTR_LANE) {
        double wx = g->qx_ - g->nx_, wy = g->qy_ - g->ny_;
        double nrm = sqrt(wx * wx + wy * wy);
        xb[0] += yh[0] * wy / nrm;
        xb[1] += -yh[0] * wx / nrm;
    }
}

static void stage_D(const orc_config *c, int i, double *lb, double *ub)
{
    if (c->constr_mode == ORC_CONSTR_LANE) { *lb = -c->lane_halfwidth; *ub = c->lane_halfwidth; }
    else { *lb = c->D_lb[i]; *ub = c->D_ub[i]; }
}

void orc_constraints(const orc_config *c, const double *x0, const double *cl, const double *U,
                     double *gout)
{
    const int nx = orc_nx(c), sm = stage_m(c);
    double X[64 * ORC_MAXNX];
    orc_rollout(c, x0, U, X);
    for (int n = 0; n < c->N; n++) {
        geom3 g;
        geom_at(c, X + n * nx, cl, &g);
        stage_constr(c, &g, X + n * nx, gout + n * sm);
    }
}

/* a-9: psi(U) = f(U) + 1/2 dist_Sigma^2(g(U) + Sigma^-1 y, D), grad by hand adjoint.
 * f: main.py:33-40 (sum of L(x_{n+1}, u_n)).  yhat out: Sigma (zeta - Pi_D zeta). */
double orc_psi(const orc_config *c, const double *x0, const double *cl, const double *U,
               const double *y, const double *Sigma, double *grad, double *yhat_out)
{
    const int nx = orc_nx(c), N = c->N, sm = stage_m(c);
    double X[(64 + 1) * ORC_MAXNX]; /* X[0] = x0, X[n+1] = x_{n+1} */
    geom3 G[64];
    double yh[64 * ORC_MAXNX];
    memcpy(X, x0, nx * sizeof(double));
    double psi = 0.0;
    for (int n = 0; n < N; n++) {
        double *xn = X + (size_t)(n + 1) * nx;
        orc_fd(c, X + (size_t)n * nx, U + 2 * n, xn);
        geom_at(c, xn, cl, &G[n]);
        psi += stage_cost_g(c, &G[n], xn, U + 2 * n, NULL, NULL);
        if (sm) {
            double gv[ORC_MAXNX];
            stage_constr(c, &G[n], xn, gv);
            for (int i = 0; i < sm; i++) {
                int k = n * sm + i;
                double lb, ub;
                stage_D(c, i, &lb, &ub);
                double zeta = gv[i] + y[k] / Sigma[k];
                double zhat = fmax(lb, fmin(zeta, ub));
                double d = zeta - zhat;
                yh[k] = Sigma[k] * d;
                psi += 0.5 * d * yh[k];
            }
        }
    }
    if (yhat_out && sm) memcpy(yhat_out, yh, (size_t)N * sm * sizeof(double));
    if (grad) {
        double lam[ORC_MAXNX] = {0};
        for (int n = N - 1; n >= 0; n--) {
            const double *xn = X + (size_t)(n + 1) * nx;
            double ub[2] = {0, 0};
            stage_cost_g(c, &G[n], xn, U + 2 * n, lam, ub);
            if (sm) stage_constr_vjp(c, &G[n], xn, yh + n * sm, lam);
            fd_vjp(c, X + (size_t)n * nx, U + 2 * n, lam, ub);
            grad[2 * n] = ub[0];
            grad[2 * n + 1] = ub[1];
        }
    }
    return psi;
}

/* --------------------------------------------------------- solver (alpaqa) */

typedef struct {
    const orc_config *c;
    const double *x0, *cl;
    int n, m;
    long n_evals; /* psi/grad evaluations (statistics) */
    int in_ls;    /* diagnostics: inside a line-search trial */
    int outer, base_iters; /* diagnostics: ALM outer iteration, inner iterations of the earlier inner solves */
    uint64_t jit;          /* evaluation jitter (orc_set_eval_jitter): generator state, 0 = off */
} prob_t;

static double dot(const double *a, const double *b, int n)
{
    double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s;
}

/* alpaqa detail::calc_x_hat / projected_gradient_step */
static void calc_xhat(const prob_t *P, double gamma, const double *x, const double *g, double *xh,
                      double *p)
{
    for (int i = 0; i < P->n; i++) {
        double lb = P->c->u_lb[i & 1], ub = P->c->u_ub[i & 1];
        /* comparison-selects keep a NaN gradient a NaN step (as Eigen's cwiseMax/cwiseMin do) */
        double lo = lb - x[i], hi = ub - x[i];
        double pi = -gamma * g[i];
        pi = pi < lo ? lo : pi;
        pi = hi < pi ? hi : pi;
        p[i] = pi; xh[i] = x[i] + pi;
    }
}

typedef struct {
    int M, n, idx, full;
    double *S, *Y; /* [M][n] */
    double *alpha, *rho;
} lbfgs_t;

static void lbfgs_reset(lbfgs_t *l) { l->idx = 0; l->full = 0; }

/* alpaqa LBFGS::update + update_valid (cbfgs eps = 0) */
static int lbfgs_update(lbfgs_t *l, const double *xk, const double *xn, const double *gk,
                        const double *gn)
{
    const double min_div = sqrt(DBL_MIN);
    double ys = 0, ss = 0;
    for (int i = 0; i < l->n; i++) {
        double s = xn[i] - xk[i], y = gn[i] - gk[i];
        ys += y * s; ss += s * s;
    }
    if (!isfinite(ys) || ss < min_div || ys < min_div) return 0;
    double *S = l->S + (size_t)l->idx * l->n, *Y = l->Y + (size_t)l->idx * l->n;
    for (int i = 0; i < l->n; i++) { S[i] = xn[i] - xk[i]; Y[i] = gn[i] - gk[i]; }
    l->idx = l->idx + 1 < l->M ? l->idx + 1 : 0;
    l->full |= l->idx == 0;
    return 1;
}

/* alpaqa LBFGS::apply(q, gamma<0, J): masked two-loop, rho recomputed on J, pairs with
 * rho <= 0 skipped, H0 = s'y / y'y of the newest valid pair.  mask[i] = 1 for i in J. */
static int lbfgs_apply_masked(lbfgs_t *l, double *q, const double *mask)
{
    if (l->idx == 0 && !l->full) return 0;
    const int n = l->n;
    const int cnt = l->full ? l->M : l->idx;
    double gamma = -1.0;
    /* newest -> oldest */
    for (int t = 0; t < cnt; t++) {
        int i = l->idx - 1 - t; if (i < 0) i += l->M;
        const double *S = l->S + (size_t)i * n, *Y = l->Y + (size_t)i * n;
        double sy = 0, sq = 0, yy = 0;
        for (int j = 0; j < n; j++) {
            sy += mask[j] * S[j] * Y[j];
            sq += mask[j] * S[j] * q[j];
            yy += mask[j] * Y[j] * Y[j];
        }
        double rho = 1.0 / sy;
        l->rho[i] = rho;
        if (!(rho > 0)) { l->rho[i] = -1.0; continue; }
        double a = rho * sq;
        l->alpha[i] = a;
        for (int j = 0; j < n; j++) q[j] -= mask[j] * a * Y[j];
        if (gamma < 0) gamma = 1.0 / (rho * yy);
    }
    if (gamma < 0) return 0;
    for (int j = 0; j < n; j++) if (mask[j] != 0.0) q[j] *= gamma;
    /* oldest -> newest */
    for (int t = cnt - 1; t >= 0; t--) {
        int i = l->idx - 1 - t; if (i < 0) i += l->M;
        if (!(l->rho[i] > 0)) continue;
        const double *S = l->S + (size_t)i * n, *Y = l->Y + (size_t)i * n;
        double yq = 0;
        for (int j = 0; j < n; j++) yq += mask[j] * Y[j] * q[j];
        double b = l->rho[i] * yq;
        double ab = l->alpha[i] - b;
        for (int j = 0; j < n; j++) q[j] += mask[j] * ab * S[j];
    }
    return 1;
}

typedef struct { int status; int iters; double eps; double psi_hat; int wrote; } inner_stats;

/* diagnostics of the last orc_solve on this thread (orc_last_ls_counters): where the evaluations of a solve
 * go -- [0] line-search trials, [1] iterations, [2] descent-lemma doublings inside line-search trials,
 * [3] descent-lemma doublings at the top of an iteration, [4] longest run of doublings in one call,
 * [5] Hessian-vector evaluations, [6] iterations whose line search fell back to tau < tau_min */
static __thread double g_lsc[8];
void orc_last_ls_counters(double *out) { memcpy(out, g_lsc, sizeof g_lsc); }

/* optional per-INNER-iteration trace (orc_solve_itertrace; study aid of the HIP-vs-oracle divergence diagnosis,
 * tools/dev/first_divergence.py): one row of ORC_ITRACE_COLS doubles per accepted iteration, holding the state
 * the iteration leaves and, for every comparison that steers the algorithm, the smallest margin by which it
 * was decided in this iteration -- a margin at rounding level is a decision another correct implementation
 * may take the other way. */
static __thread double *g_it = NULL;
static __thread int g_it_rows = 0, g_it_n = 0;
static __thread double g_it_dl, g_it_ls, g_it_act, g_it_heur;
static void it_min(double *slot, double v) { v = fabs(v); if (v < *slot) *slot = v; }

/* Study aid (orc_set_eval_jitter): every psi and every gradient component the solver sees is moved by a random
 * whole number of ulps in [-ulps, ulps] -- what ANOTHER correct implementation of the same formulae (other
 * summation order, other libm, fused multiply-adds) would hand the same algorithm.  Used to measure how far the
 * solver's path and end point depend on the last bits of its evaluations: the yardstick the HIP-vs-oracle
 * agreement is held against (tests/test_oracle_golden.py, tests/test_gpu_parity.py).  Process-wide, off by default. */
static int g_jitter_ulps = 0;
static uint64_t g_jitter_seed = 0;
void orc_set_eval_jitter(int ulps, uint64_t seed) { g_jitter_ulps = ulps < 0 ? 0 : ulps; g_jitter_seed = seed; }
static double jitter(prob_t *P, double v)
{
    P->jit = P->jit * 6364136223846793005ULL + 1442695040888963407ULL;
    const int k = (int)((P->jit >> 33) % (uint64_t)(2 * g_jitter_ulps + 1)) - g_jitter_ulps;
    if (k == 0 || !isfinite(v) || v == 0.0) return v;
    int64_t b; memcpy(&b, &v, 8); b += k; memcpy(&v, &b, 8);   /* k ulps along the number line of |v| */
    return v;
}

/* study aid (orc_solve_dump_evals): every point evaluated during ONE inner iteration, with the value returned */
static __thread double *g_dump = NULL;
static __thread int g_dump_rows = 0, g_dump_n = 0, g_dump_iter = -1, g_dump_on = 0;

static double eval_psi(prob_t *P, const double *x, const double *y, const double *Sig, double *grad,
                       double *yhat)
{
    P->n_evals++;
    double v = orc_psi(P->c, P->x0, P->cl, x, y, Sig, grad, yhat);
    if (g_dump_on && g_dump_n < g_dump_rows) {
        double *t = g_dump + (size_t)g_dump_n++ * (P->n + 2);
        t[0] = grad ? 1.0 : 0.0; t[1] = v; memcpy(t + 2, x, P->n * sizeof(double));
    }
    if (P->jit) {
        v = jitter(P, v);
        if (grad) for (int i = 0; i < P->n; i++) grad[i] = jitter(P, grad[i]);
    }
    return v;
}

/* alpaqa detail::descent_lemma */
static void descent_lemma(prob_t *P, const double *y, const double *Sig, const double *xk,
                          double psik, const double *gk, double *xh, double *p, double *yhx,
                          double *psixh, double *pp, double *gp, double *L, double *gamma)
{
    const orc_config *c = P->c;
    double margin = (1.0 + fabs(psik)) * c->qub_tol;
    int run = 0;
    for (;;) {
        if (g_it) it_min(&g_it_dl, ((*psixh - psik) - (*gp + 0.5 * (*L) * (*pp) + margin)) / (1.0 + fabs(psik)));
        if (!(*psixh - psik > *gp + 0.5 * (*L) * (*pp) + margin)) break;
        if (!((*L) * 2.0 <= c->L_max)) break;
        g_lsc[P->in_ls ? 2 : 3] += 1; if (++run > g_lsc[4]) g_lsc[4] = run;
        *L *= 2.0; *gamma /= 2.0;
        calc_xhat(P, *gamma, xk, gk, xh, p);
        *gp = dot(gk, p, P->n);
        *pp = dot(p, p, P->n);
        *psixh = eval_psi(P, xh, y, Sig, NULL, yhx);
    }
}

/* alpaqa detail::calc_augmented_lagrangian_hessian_prod_fd */
static void hess_prod_fd(prob_t *P, const double *y, const double *Sig, const double *xk,
                         const double *gk, const double *v, double *Hv, double *work)
{
    const int n = P->n;
    double h = cbrt(DBL_EPSILON) * (1.0 + sqrt(dot(xk, xk, n)));
    for (int i = 0; i < n; i++) work[i] = xk[i] + h * v[i];
    g_lsc[5] += 1;
    eval_psi(P, work, y, Sig, Hv, NULL);
    for (int i = 0; i < n; i++) Hv[i] = (Hv[i] - gk[i]) / h;
}

/* a-9..a-12: alpaqa StructuredPANOCLBFGSSolver::operator() with the parameters of
 * controller.py:27-37 (stop_crit ProjGradNorm2, L-BFGS memory N_horiz) */
static inner_stats panoc(prob_t *P, const double *Sig, double eps, int always_overwrite, int max_iter,
                         double *x, double *y, double *err_z, double *wk, lbfgs_t *lb)
{
    const orc_config *c = P->c;
    const int n = P->n, m = P->m;
    inner_stats st = {ORC_ST_UNKNOWN, 0, INFINITY, 0.0, 0};
    double *xk = wk, *xh = xk + n, *xn = xh + n, *xhn = xn + n, *p = xhn + n, *pn = p + n,
           *q = pn + n, *gk = q + n, *gn = gk + n, *HqK = gn + n, *work = HqK + n,
           *mask = work + n, *yhx = mask + n, *yhxn = yhx + (m ? m : 1);
    memcpy(xk, x, n * sizeof(double));
    lbfgs_reset(lb);
    int no_progress = 0;

    /* initial Lipschitz estimate (alpaqa detail::initial_lipschitz_estimate) */
    double hn2 = 0;
    for (int i = 0; i < n; i++) {
        double h = fmax(fabs(xk[i] * c->lip_eps), c->lip_delta);
        work[i] = xk[i] + h; hn2 += h * h;
    }
    eval_psi(P, work, y, Sig, gn, NULL);
    double psik = eval_psi(P, xk, y, Sig, gk, NULL);
    double dn2 = 0;
    for (int i = 0; i < n; i++) { double d = gn[i] - gk[i]; dn2 += d * d; }
    double Lk = sqrt(dn2) / sqrt(hn2);
    /* std::clamp: NaN stays NaN */
    Lk = Lk < c->L_min ? c->L_min : (c->L_max < Lk ? c->L_max : Lk);
    if (!isfinite(Lk) || psik != psik) { st.status = ORC_ST_NOTFINITE; return st; }
    double gamma = c->Lgamma_factor / Lk;
    double tau = NAN;

    calc_xhat(P, gamma, xk, gk, xh, p);
    double psixh = eval_psi(P, xh, y, Sig, NULL, yhx);
    double gp = dot(gk, p, n), pp = dot(p, p, n);
    double phik = psik + pp / (2.0 * gamma) + gp;

    for (int k = 0; k <= max_iter; k++) {
        g_dump_on = g_dump != NULL && P->base_iters + k + 1 == g_dump_iter;
        int gamma_changed_top = 0;
        double gamma_old_top = gamma;
        /* hessian_step_size_heuristic (controller.py:32) [RECALLED, safeguarded]:
         * Cauchy step length from an FD Hessian-vector product along grad */
        if (k > 0 && c->hess_heuristic > 0 && k % c->hess_heuristic == 0) {
            hess_prod_fd(P, y, Sig, xk, gk, gk, HqK, work);
            double gHg = dot(gk, HqK, n), gg = dot(gk, gk, n);
            double eta = gg / gHg;
            if (g_it) it_min(&g_it_heur, (eta * c->Lgamma_factor - gamma) / gamma);
            if (eta > 0 && isfinite(eta) && eta * c->Lgamma_factor > gamma) {
                Lk = 1.0 / eta;
                gamma = c->Lgamma_factor / Lk;
                calc_xhat(P, gamma, xk, gk, xh, p);
                psixh = eval_psi(P, xh, y, Sig, NULL, yhx);
                gp = dot(gk, p, n); pp = dot(p, p, n);
                gamma_changed_top = 1;
            }
        }
        if (k == 0 || gamma_changed_top) {
            descent_lemma(P, y, Sig, xk, psik, gk, xh, p, yhx, &psixh, &pp, &gp, &Lk, &gamma);
            if (k > 0 && gamma != gamma_old_top) lbfgs_reset(lb);
            phik = psik + pp / (2.0 * gamma) + gp;
        }

        /* stop criterion ProjGradNorm2 (controller.py:29): ||p|| / gamma */
        double epsk = sqrt(pp) / gamma;
        /* the stop test ON the iterate the previous row describes (same inner solve: k > 0) */
        if (g_it && k > 0 && g_it_n > 0 && g_it_n <= g_it_rows) g_it[(size_t)(g_it_n - 1) * ORC_ITRACE_COLS + 18] = epsk / eps - 1.0;
        int stop = epsk <= eps ? ORC_ST_CONVERGED
                 : (c->max_total_evals > 0 && P->n_evals >= c->max_total_evals) ? ORC_ST_MAXTIME
                 : k == max_iter ? ORC_ST_MAXITER
                 : !isfinite(epsk) ? ORC_ST_NOTFINITE
                 : no_progress > c->max_no_progress ? ORC_ST_NOPROGRESS : ORC_ST_UNKNOWN;
        if (stop != ORC_ST_UNKNOWN) {
            /* (deviation, for the caller's safety: a NotFinite inner solve never hands back its
             * iterate -- alpaqa would move a NaN x-hat into x when always_overwrite is set) */
            if (stop == ORC_ST_CONVERGED || (always_overwrite && stop != ORC_ST_NOTFINITE)) {
                if (m) {
                    /* calc_err_z: g(xh) - Pi_D(g(xh) + Sigma^-1 y) */
                    orc_constraints(c, P->x0, P->cl, xh, err_z);
                    int sm = stage_m(c);
                    for (int i = 0; i < m; i++) {
                        double lbd, ubd; stage_D(c, i % sm, &lbd, &ubd);
                        double z = err_z[i] + y[i] / Sig[i];
                        err_z[i] = err_z[i] - fmax(lbd, fmin(z, ubd));
                    }
                    memcpy(y, yhx, m * sizeof(double));
                }
                memcpy(x, xh, n * sizeof(double));
                st.psi_hat = psixh; st.wrote = 1;
            }
            st.status = stop; st.iters = k; st.eps = epsk;
            return st;
        }

        /* structured quasi-Newton direction (a-11) */
        int nJ = 0;
        if (k > 0) {
            for (int i = 0; i < n; i++) {
                double lbi = c->u_lb[i & 1], ubi = c->u_ub[i & 1];
                double gd = xk[i] - gamma * gk[i];
                if (g_it) { it_min(&g_it_act, gd - lbi); it_min(&g_it_act, ubi - gd); }
                if (gd < lbi || ubi < gd) { q[i] = p[i]; mask[i] = 0.0; }
                else { q[i] = 0.0; mask[i] = 1.0; nJ++; }
            }
            if (nJ > 0) {
                if (nJ == n) {
                    for (int i = 0; i < n; i++) q[i] = -gk[i];
                } else {
                    hess_prod_fd(P, y, Sig, xk, gk, q, HqK, work);
                    for (int i = 0; i < n; i++) if (mask[i] != 0.0) q[i] = -gk[i] - HqK[i];
                }
                int ok = lbfgs_apply_masked(lb, q, mask);
                if (!ok) for (int i = 0; i < n; i++) if (mask[i] != 0.0) q[i] *= gamma;
            }
        }

        /* line search on the forward-backward envelope (a-12) */
        tau = 1.0;
        double sig_pp = (1.0 - gamma * Lk) * pp / (2.0 * gamma);
        double margin = (1.0 + fabs(phik)) * c->qub_tol;
        if (k == 0) tau = 0.0;
        else {
            int fin = 1;
            for (int i = 0; i < n; i++) if (!isfinite(q[i])) fin = 0;
            if (!fin) { tau = 0.0; lbfgs_reset(lb); }
            else if (nJ == 0) tau = 0.0;
        }
        double phin, psin, psixhn, gpn, ppn, Ln, gamman, ls_cond;
        g_lsc[1] += 1; P->in_ls = 1;
        int it_trials = 0; double qub_before = g_lsc[2];
        double tau_used = tau; int safe_step = 0;
        do {
            g_lsc[0] += 1; it_trials++;
            tau_used = tau; safe_step = tau / 2.0 < c->tau_min;
            Ln = Lk; gamman = gamma;
            if (tau / 2.0 < c->tau_min) {
                g_lsc[6] += 1;
                memcpy(xn, xh, n * sizeof(double));
                psin = psixh;
                eval_psi(P, xn, y, Sig, gn, NULL); /* calc_grad_psi_from_yhat */
            } else {
                if (tau == 1.0) for (int i = 0; i < n; i++) xn[i] = xk[i] + q[i];
                else for (int i = 0; i < n; i++) xn[i] = xk[i] + (1.0 - tau) * p[i] + tau * q[i];
                psin = eval_psi(P, xn, y, Sig, gn, NULL);
            }
            calc_xhat(P, gamman, xn, gn, xhn, pn);
            psixhn = eval_psi(P, xhn, y, Sig, NULL, yhxn);
            gpn = dot(gn, pn, n); ppn = dot(pn, pn, n);
            /* update_lipschitz_in_linesearch = true */
            descent_lemma(P, y, Sig, xn, psin, gn, xhn, pn, yhxn, &psixhn, &ppn, &gpn, &Ln, &gamman);
            phin = psin + ppn / (2.0 * gamman) + gpn;
            ls_cond = phin - (phik - sig_pp);
            if (g_it) it_min(&g_it_ls, (ls_cond - margin) / (1.0 + fabs(phik)));
            tau /= 2.0;
            /* a NaN condition (the trial point's evaluation overflowed) is a failed trial, like +inf:
             * alpaqa's literal `ls_cond > margin` would accept it because NaN compares false */
        } while (!(ls_cond <= margin) && tau >= c->tau_min);
        P->in_ls = 0;
        /* dependent evaluation trips of this line search if up to four trial points are evaluated side by side:
         * one trip for the gradients of up to four trials, one per two trials for their costs (+ speculative
         * gradients), one per descent-lemma doubling */
        g_lsc[7] += (it_trials + 3) / 4 + (it_trials + 1) / 2 + (g_lsc[2] - qub_before);

        if (gamma != gamman) lbfgs_reset(lb);
        const int pair_ok = lbfgs_update(lb, xk, xn, gk, gn);

        if (no_progress > 0 || k % c->max_no_progress == 0) {
            int same = 1;
            for (int i = 0; i < n; i++) if (xk[i] != xn[i]) { same = 0; break; }
            no_progress = same ? no_progress + 1 : 0;
        }

        Lk = Ln; gamma = gamman; psik = psin; psixh = psixhn; phik = phin;
        memcpy(xk, xn, n * sizeof(double));
        memcpy(xh, xhn, n * sizeof(double));
        memcpy(p, pn, n * sizeof(double));
        memcpy(gk, gn, n * sizeof(double));
        if (m) memcpy(yhx, yhxn, m * sizeof(double));
        gp = gpn; pp = ppn;
        if (g_it) {
            if (g_it_n < g_it_rows) {
                double *t = g_it + (size_t)g_it_n * ORC_ITRACE_COLS;
                /* tau of the accepted trial (the HIP record holds it halved, as `tau` is here by now); the safe prox
                 * step x+ = xhat is reported as its negative */
                t[0] = P->base_iters + k + 1; t[1] = P->outer; t[2] = k + 1; t[3] = eps;
                t[4] = safe_step ? -tau_used : tau_used;
                t[5] = it_trials; t[6] = Lk; t[7] = gamma; t[8] = nJ; t[9] = lb->full ? lb->M : lb->idx;
                t[10] = pair_ok; t[11] = psik; t[12] = phik; t[13] = pp; t[14] = (double)P->n_evals;
                t[15] = g_it_ls; t[16] = g_it_dl; t[17] = g_it_act; t[18] = NAN; t[19] = g_it_heur;
            }
            g_it_n++;
            g_it_ls = g_it_dl = g_it_act = g_it_heur = INFINITY;
        }
    }
    st.status = ORC_ST_MAXITER; st.iters = max_iter;
    return st;
}

/* alpaqa detail::update_penalty_weights (single_penalty_factor = false) */
static void update_penalty(const orc_config *c, double Delta, int first, const double *e1,
                           const double *e2, double ne1, const double *Sig_old, double *Sig, int m)
{
    if (ne1 <= c->alm_delta) { memcpy(Sig, Sig_old, m * sizeof(double)); return; }
    for (int i = 0; i < m; i++) {
        if (first || fabs(e1[i]) > c->theta * fabs(e2[i]))
            Sig[i] = fmin(c->Sigma_max, fmax(Delta * fabs(e1[i]) / ne1, 1.0) * Sig_old[i]);
        else
            Sig[i] = Sig_old[i];
    }
}

static double norm_inf(const double *v, int m)
{
    double r = 0;
    for (int i = 0; i < m; i++) { double a = fabs(v[i]); if (a > r || a != a) r = a; if (r != r) break; }
    return r; /* NaN-propagating */
}

/* a-8: alpaqa ALMSolver::operator() with controller.py:39-48 parameters; wall-clock
 * caps (controller.py:30,:44) replaced by iteration caps (deliberate deviation). */
/* optional per-outer-iteration trace (test / study aid): rows of ORC_TRACE_COLS doubles */
static __thread double *g_trace = NULL;
static __thread int g_trace_rows = 0, g_trace_n = 0;
int orc_solve_traced(const orc_config *c, const double *x0, const double *cl, double *U, double *lam,
                     double *stats, double *trace, int max_rows)
{
    g_trace = trace; g_trace_rows = max_rows; g_trace_n = 0;
    orc_solve(c, x0, cl, U, lam, stats);
    g_trace = NULL;
    return g_trace_n;
}

int orc_solve_dump_evals(const orc_config *c, const double *x0, const double *cl, double *U, double *lam,
                         double *stats, int iteration, double *out, int max_rows)
{
    g_dump = out; g_dump_rows = max_rows; g_dump_n = 0; g_dump_iter = iteration; g_dump_on = 0;
    orc_solve(c, x0, cl, U, lam, stats);
    g_dump = NULL; g_dump_on = 0;
    return g_dump_n;
}

int orc_solve_itertrace(const orc_config *c, const double *x0, const double *cl, double *U, double *lam,
                        double *stats, double *trace, int max_rows)
{
    g_it = trace; g_it_rows = max_rows; g_it_n = 0;
    g_it_ls = g_it_dl = g_it_act = g_it_heur = INFINITY;
    orc_solve(c, x0, cl, U, lam, stats);
    g_it = NULL;
    return g_it_n;
}

void orc_solve(const orc_config *c, const double *x0, const double *cl, double *U, double *lam,
               double *stats)
{
    prob_t P = {c, x0, cl, 2 * c->N, orc_m(c), 0, 0, 0, 0, 0};
    if (g_jitter_ulps > 0) {   /* a generator per solve, keyed by the agent's initial state: the same jitter whatever the thread */
        uint64_t k0; memcpy(&k0, &x0[0], 8);
        uint64_t k1; memcpy(&k1, &x0[1], 8);
        P.jit = (k0 * 0x9E3779B97F4A7C15ULL) ^ (k1 + g_jitter_seed * 0xD1B54A32D192ED03ULL) ^ 0x2545F4914F6CDD1DULL;
        if (P.jit == 0) P.jit = 1;
    }
    memset(g_lsc, 0, sizeof g_lsc);
    const int n = P.n, m = P.m, mm = m ? m : 1;
    const int M = c->lbfgs_memory;
    double *wk = (double *)malloc(sizeof(double) * (12 * (size_t)n + 2 * mm));
    double *Sig = (double *)malloc(sizeof(double) * 5 * mm);
    double *Sig_old = Sig + mm, *e1 = Sig_old + mm, *e2 = e1 + mm, *ysave = e2 + mm;
    lbfgs_t lb;
    lb.M = M; lb.n = n; lb.idx = 0; lb.full = 0;
    lb.S = (double *)malloc(sizeof(double) * (2 * (size_t)M * n + 2 * M));
    lb.Y = lb.S + (size_t)M * n; lb.alpha = lb.Y + (size_t)M * n; lb.rho = lb.alpha + M;
    (void)ysave;

    for (int i = 0; i < m; i++) { Sig[i] = c->Sigma0; Sig_old[i] = NAN; e1[i] = NAN; e2[i] = NAN; }
    double ne1 = NAN, ne2 = NAN;
    double eps = c->eps0, eps_old = NAN, Delta = c->Delta, rho = c->rho;
    int first = 1, init_red = 0, pen_red = 0;
    int status = ORC_ST_UNKNOWN, outer = 0, inner_it = 0, inner_fail = 0;
    double out_eps = INFINITY, out_delta = INFINITY, out_psi = 0.0;
    int sm = stage_m(c);

    for (int i = 0; i < c->max_outer; i++) {
        /* detail::project_y */
        for (int k = 0; k < m; k++) {
            double lbd, ubd; stage_D(c, k % sm, &lbd, &ubd);
            double ylo = isinf(lbd) ? 0.0 : -c->M, yhi = isinf(ubd) ? 0.0 : c->M;
            lam[k] = fmin(fmax(lam[k], ylo), yhi);
        }
        int out_of_pen = (first ? init_red == c->max_num_initial_retries
                                : pen_red == c->max_num_retries) ||
                         (init_red + pen_red == c->max_total_num_retries);
        int out_of_iter = i + 1 == c->max_outer;
        int budget = c->max_total_inner - inner_it;
        int max_it = c->max_iter < budget ? c->max_iter : budget;
        /* the last inner solve the budget allows always hands back its iterate */
        int last_by_budget = max_it >= budget;
        int overwrite = out_of_iter || out_of_pen || last_by_budget;
        P.outer = i; P.base_iters = inner_it;
        inner_stats ps = panoc(&P, Sig, eps, overwrite, max_it, U, lam, e2, wk, &lb);
        int conv = ps.status == ORC_ST_CONVERGED;
        if (ps.wrote) out_psi = ps.psi_hat;
        inner_fail += !conv;
        inner_it += ps.iters;
        if (ps.status == ORC_ST_NOTFINITE && ps.iters == 0) {
            /* psi or its gradient is non-finite AT the point the inner solve starts from: no penalty or
             * tolerance change repairs that, the solve ends with the inner status (deliberate: alpaqa
             * 0.0.1 would walk through its retries, two evaluations each, and end as MaxIter after
             * max_outer of them; controller.py:64 counts either as a failure).  A NotFinite that shows
             * up LATER in an inner solve (a line-search trial that overflowed and was accepted because
             * NaN compares false, as in alpaqa) takes the ordinary not-converged path below. */
            out_eps = ps.eps; out_delta = ne1; outer = i + 1; status = ORC_ST_NOTFINITE;
            break;
        }
        int out_of_time = inner_it >= c->max_total_inner ||
                          (c->max_total_evals > 0 && P.n_evals >= c->max_total_evals);
        int backtrack = !conv && !overwrite && !out_of_time;
        if (g_trace && g_trace_n < g_trace_rows) {
            double *t = g_trace + (size_t)g_trace_n++ * ORC_TRACE_COLS;
            double smin = INFINITY, smax = 0;
            for (int k = 0; k < m; k++) { smin = fmin(smin, Sig[k]); smax = fmax(smax, Sig[k]); }
            t[0] = i; t[1] = eps; t[2] = ps.status; t[3] = ps.iters; t[4] = ps.eps;
            t[5] = ps.wrote ? norm_inf(e2, m) : NAN; t[6] = smin; t[7] = smax; t[8] = backtrack;
            t[9] = overwrite; t[10] = (double)P.n_evals; t[11] = norm_inf(lam, m);
        }
        if (backtrack) {
            if (!first) {
                Delta = fmax(1.0, Delta * c->Delta_lower);
                update_penalty(c, Delta, first, e1, e2, ne1, Sig_old, Sig, m);
                rho = fmin(0.5, rho * c->rho_increase);
                eps = fmax(rho * eps_old, c->alm_eps);
                pen_red++;
            } else {
                for (int k = 0; k < m; k++) Sig[k] *= c->Sigma0_lower;
                eps *= c->eps0_increase;
                init_red++;
            }
        } else {
            /* error2.swap(error1) */
            for (int k = 0; k < m; k++) { double t = e1[k]; e1[k] = e2[k]; e2[k] = t; }
            ne2 = ne1; ne1 = norm_inf(e1, m);
            int alm_conv = ps.eps <= c->alm_eps && conv && ne1 <= c->alm_delta;
            if (alm_conv || out_of_iter || out_of_time) {
                out_eps = ps.eps; out_delta = ne1; outer = i + 1;
                status = alm_conv ? ORC_ST_CONVERGED : out_of_time ? ORC_ST_MAXTIME : ORC_ST_MAXITER;
                break;
            }
            for (int k = 0; k < m; k++) { double t = Sig_old[k]; Sig_old[k] = Sig[k]; Sig[k] = t; }
            update_penalty(c, Delta, first, e1, e2, ne1, Sig_old, Sig, m);
            eps_old = eps; eps = fmax(rho * eps, c->alm_eps);
            first = 0;
        }
        outer = i + 1;
    }
    (void)ne2;
    stats[0] = status; stats[1] = outer; stats[2] = inner_it; stats[3] = inner_fail;
    stats[4] = out_eps; stats[5] = out_delta;
    stats[6] = out_psi; /* psi(xhat) of the last inner solve that handed back its iterate */
    stats[7] = (double)P.n_evals;
    free(lb.S); free(Sig); free(wk);
}

/* ------------------------------------------------------------------ batch */

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_solve_batch(const orc_config *c, int B, const double *x0, const double *cl,
                     const int32_t *cl_index, double *U, double *lam, double *stats, int nthreads)
{
    const int nx = orc_nx(c), n = 2 * c->N, m = orc_m(c);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int b = 0; b < B; b++) {
        const double *clb = cl + (size_t)(cl_index ? cl_index[b] : 0) * 2 * c->S;
        double dummy = 0;
        orc_solve(c, x0 + (size_t)b * nx, clb, U + (size_t)b * n, m ? lam + (size_t)b * m : &dummy,
                  stats + (size_t)b * ORC_NSTATS);
    }
}

void orc_psi_batch(const orc_config *c, int B, const double *x0, const double *cl,
                   const int32_t *cl_index, const double *U, const double *y, const double *Sigma,
                   double *psi, double *grad, int nthreads)
{
    const int nx = orc_nx(c), n = 2 * c->N, m = orc_m(c);
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int b = 0; b < B; b++) {
        const double *clb = cl + (size_t)(cl_index ? cl_index[b] : 0) * 2 * c->S;
        psi[b] = orc_psi(c, x0 + (size_t)b * nx, clb, U + (size_t)b * n,
                         m ? y + (size_t)b * m : NULL, m ? Sigma + (size_t)b * m : NULL,
                         grad ? grad + (size_t)b * n : NULL, NULL);
    }
}

/* ------------------------------------------------- f-3: game_theory.py lane-change payoffs */

typedef struct { double x, v; int lane; } gcar;
typedef struct { double L, W, l, th, tlc, td, ti, tau, amax, h, Lf, q1, q2, a, b; } gpar;

/* game_theory.py:115-153 Car.get_safety_distance */
static double g_safety_distance(const gpar *p, const gcar *s, const gcar *c, int target_lane)
{
    const double dv = s->v - c->v;
    if (s->lane == c->lane) {
        if (s->x > c->x) return fabs(s->x - c->x);
        if (target_lane == s->lane)
            return p->q1 * s->v + p->td + p->q2 * (dv * p->tau + p->ti / 2 + dv * dv / (2 * p->amax)) + p->l;
        if (s->v > c->v) return s->v - c->v * p->tlc / 2 + p->L + p->W / 2 * sin(p->th); /* S01 */
        return p->q1 * s->v * p->td + p->l;
    }
    if (s->x < c->x) { /* S02 */
        if (s->v > c->v)
            return s->v - c->v * p->tlc / 2 + p->L - p->W / 2 * sin(p->th) + p->q1 * s->v * p->td +
                   p->q2 * (dv * p->tau + p->ti / 2 + dv * dv / (2 * p->amax));
        return p->q1 * s->v * p->td + p->l;
    }
    if (s->v < c->v) { /* S03 */
        const double du = c->v - s->v;
        return du * 3 / 4 * p->tlc + p->L + p->q1 * c->v * p->td +
               p->q2 * (du * p->tau + p->ti / 2 + du * du / (2 * p->amax));
    }
    return p->q1 * c->v * p->td + p->l;
}

/* game_theory.py:155-177 Car.get_safety_payoff */
static double g_safety_payoff(const gpar *p, const gcar *s, const gcar *cars, int n, int target_lane)
{
    double payoff = 1, temp = 1;
    for (int i = 0; i < n; i++) {
        const gcar *c = &cars[i];
        if (s->lane != c->lane && s->lane == target_lane) continue;
        const double Sk = g_safety_distance(p, s, c, target_lane);
        const double Dk = fabs(s->x - c->x);
        if (Dk >= fabs(Sk)) temp = 1;
        if (Dk <= p->l) temp = -1;
        if (p->l < Dk && Dk < fabs(Sk)) temp = log(Dk / Sk + 1) / log(2.0);
        if (temp < payoff) payoff = temp;
    }
    return payoff;
}

/* game_theory.py:61-75 Car.get_car_in_front: nearest car ahead in the target lane (first minimum) */
static int g_car_in_front(const gcar *s, const gcar *cars, int n, int target_lane)
{
    int best = -1;
    for (int i = 0; i < n; i++) {
        if (cars[i].lane != target_lane) continue;
        if (cars[i].x > s->x) {
            if (best < 0) best = i;
            if (cars[best].x > cars[i].x) best = i;
        }
    }
    return best;
}

/* game_theory.py:77-90 Car.get_car_behind: nearest car behind in lane 2 */
static int g_car_behind(const gcar *s, const gcar *cars, int n)
{
    int best = -1;
    for (int i = 0; i < n; i++) {
        if (cars[i].lane != 2) continue;
        if (cars[i].x < s->x) {
            if (best < 0) best = i;
            if (cars[best].x < cars[i].x) best = i;
        }
    }
    return best;
}

/* game_theory.py:179-190 */
static double g_velocity_payoff(const gcar *s, const gcar *cars, int n, int target_lane)
{
    const int f = g_car_in_front(s, cars, n, target_lane);
    if (f < 0) return 1;
    if (cars[f].v == 0) return -1;
    if (cars[f].v >= 2 * s->v) return 1;
    return (cars[f].v - s->v) / s->v;
}

/* game_theory.py:192-203 (+ :92-113 for tca) */
static double g_comfort_payoff(const gpar *p, const gcar *s, const gcar *cars, int n, int target_lane)
{
    if (target_lane == 1) return 0;
    const int f = g_car_in_front(s, cars, n, 1);
    if (f < 0) return 0;
    if (s->v > cars[f].v) {
        const double Li = p->Lf + p->l;
        const double Di = Li * cos(atan2(p->W, 2 * p->Lf) - p->th);
        const double D1 = cars[f].x - s->x;
        const double tc1 = D1 / (s->v - cars[f].v);
        const double Px2 = s->v * tc1 - Di;
        const double tca = Px2 / (s->v - cars[f].v);
        return 2 / (1 + exp(-tca)) - 2;
    }
    return 0;
}

/* game_theory.py:205-244 Car.get_total_payoff (the module-global `ego` at :228 is the caller) */
static double g_total_payoff(const gpar *p, const gcar *ego, const gcar *cars, int n, int target_lane,
                             double *safety, double *velocity)
{
    *safety = g_safety_payoff(p, ego, cars, n, target_lane);
    *velocity = g_velocity_payoff(ego, cars, n, target_lane);
    const double total = p->a * *safety + p->b * *velocity;
    const int bi = g_car_behind(ego, cars, n);
    double total_behind = 0;
    if (bi >= 0) {
        gcar others[64];
        int m = 0;
        for (int i = 0; i < n; i++) if (i != bi) others[m++] = cars[i];
        if (target_lane == 2) { others[m].x = ego->x; others[m].v = ego->v; others[m].lane = 2; m++; }
        const double sb = g_safety_payoff(p, &cars[bi], others, m, 2);
        const double vb = g_velocity_payoff(&cars[bi], others, m, 2);
        total_behind = p->a * sb + p->b * vb;
    }
    return total + total_behind;
}

void orc_lane_payoff(const double *params, int B, int K, const double *ego, const double *cars,
                     const int32_t *ncars, double *out)
{
    gpar p;
    memcpy(&p, params, sizeof p);
    for (int b = 0; b < B; b++) {
        gcar e = {ego[3 * b], ego[3 * b + 1], (int)ego[3 * b + 2]};
        gcar cs[63];
        const int n = ncars[b];
        for (int i = 0; i < n; i++) {
            const double *c = cars + ((size_t)b * K + i) * 3;
            cs[i].x = c[0]; cs[i].v = c[1]; cs[i].lane = (int)c[2];
        }
        for (int t = 1; t <= 2; t++) {
            double *o = out + ((size_t)b * 2 + (t - 1)) * 4;
            o[0] = g_total_payoff(&p, &e, cs, n, t, &o[1], &o[2]);
            o[3] = g_comfort_payoff(&p, &e, cs, n, t);
        }
    }
}
