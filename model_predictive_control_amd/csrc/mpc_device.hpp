// mpc_device.hpp -- gfx950 device code of the MPC hot path: bicycle models, RK4 stage,
// tracking errors, stage cost and their hand-derived adjoints.
//
// Reference behaviour restated (paths into the upstream repo):
//   car_dynamics.py:93-129   Pacejka bicycle RHS (nx = 6)
//   dynamics.py:144-173      kinematic bicycle RHS (nx = 4)
//   car_dynamics.py:136-145  cs.integrator("rk"): nfe classical RK4 steps per stage
//   car_dynamics.py:174-228  nearest centerline point, cte / heading / pos errors
//   car_dynamics.py:230-258  stage cost
//   main.py:33-52            objective and per-stage constraints
// One thread owns one agent; everything is fp64 VALU work (no dense contraction, so no MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpc {

constexpr int KIN = 0;
constexpr int PAC = 1;

// Flattened, device-friendly copy of mpc_config (passed by value as a kernel argument, so every
// field is read through scalar loads).
struct DevCfg {
    int model, N, S, nfe, wrap_mode, clip_inputs, constr_mode, sm; // sm = constraints per stage
    int nx, n, m, M;                                                // n = 2N, m = sm*N, M = L-BFGS memory
    int max_iter, max_outer, hess_heuristic, max_no_progress;
    int max_num_initial_retries, max_num_retries, max_total_num_retries, max_total_inner, max_total_evals;
    int no_spec;         // MPC_NO_SPEC: no speculative gradients (same results, more rounds)
    int chain;           // agents that wait in PH_W_LS_G are served by one THREAD each, in extra workgroups of the
                         // step-kernel launch (chain_block; n <= 64; the host sets it per launch: full rounds of big
                         // groups only; MPC_NO_CHAIN: never -- same results)
    int all_rows;        // MPC_ALL_ROWS: the step kernel fetches all six rows of an agent whatever its phase (same results)
    int no_memo;         // MPC_NO_MEMO / mpc_set_memo(h, 0): failed retries are recomputed, not replayed (same results)
    int no_la;           // MPC_NO_LOOKAHEAD: the persistent kernel evaluates only what the state machine asks for (same results)
    double h;      // RK4 step Ts / nfe
    double v_ref;
    double w[6];
    double lf, lr, mass, inv_mass, inv_iz;
    double bf, cf, df, br, cr, dr, cm1, cm2, cr0, cr2;
    double accel, friction, max_drive, max_steer;
    double u_lb[2], u_ub[2];
    double g_off[6], D_lb[6], D_ub[6], lane_hw;
    double alm_eps, alm_delta, Sigma0, eps0, rho, Delta, theta, Mcap, Sigma_max;
    double Delta_lower, Sigma0_lower, eps0_increase, rho_increase;
    double lip_eps, lip_delta, Lgamma, L_min, L_max, tau_min, qub_tol;
};

#define MPC_DEV __device__ __forceinline__

// ---------------------------------------------------------------------------------- math
// The OCML double-precision transcendentals are full-range (Payne-Hanek reduction, dozens of
// 64-bit literals each) and dominate this kernel's instruction count.  The angles of this problem
// are O(1), so the hot path uses lean kernels: 3-term Cody-Waite reduction with FMA + the classic
// minimax polynomials on [-pi/4, pi/4] (sin/cos) and on [-0.4143, 0.4143] with a three-interval
// reduction (atan/atan2); <= 2 ulp against libm (tests/test_gpu_parity.py::test_device_math).
// Arguments outside the fast range fall back to OCML, out of line.
struct SinCos { double s, c; };

__device__ __noinline__ SinCos ocml_sincos(double x) { SinCos r; ::sincos(x, &r.s, &r.c); return r; }
__device__ __noinline__ double m_tan(double x) { return ::tan(x); }
__device__ __noinline__ double ocml_fmod(double a, double b) { return ::fmod(a, b); }
__device__ __noinline__ double ocml_atan2(double y, double x) { return ::atan2(y, x); }
__device__ __noinline__ double ocml_remainder(double a, double b) { return ::remainder(a, b); }

#ifdef MPC_ATAN_OUTLINE
#define MPC_ATAN_FN __device__ __noinline__
#else
#define MPC_ATAN_FN __device__ __forceinline__ // measured faster inlined for both models (profiles/)
#endif

// sin and cos of a reduced argument |r| <= pi/4 (the minimax kernels, no reduction, no quadrant)
__device__ __forceinline__ SinCos kernel_sincos(double r)
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    const double z = r * r;
    // sin kernel
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    SinCos o;
    o.s = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
    // cos kernel
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double wv = 1.0 - hz;
    o.c = wv + (((1.0 - wv) - hz) + z * z * pc);
    return o;
}

__device__ __forceinline__ SinCos lean_sincos(double x) // |x| < 1e5
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    const double kf = rint(x * 6.36619772367581382433e-01); // 2/pi
    double r = fma(-kf, 1.57079632679489655800e+00, x);
    r = fma(-kf, 6.12323399573676603587e-17, r);
    r = fma(-kf, -1.49738490485916983294e-33, r);
    const int q = (int)kf;
    const SinCos k = kernel_sincos(r);
    const double sn = k.s, cs = k.c;
    SinCos o;
    const bool swap = q & 1;
    const double sv = swap ? cs : sn, cv = swap ? sn : cs;
    o.s = (q & 2) ? -sv : sv;
    o.c = ((q + 1) & 2) ? -cv : cv;
    return o;
}

// sin/cos(a + d) from sin/cos(a) and a small increment |d| <= pi/4: one rotation instead of a
// reduction + quadrant selection (~2 ulp of 1 more than a direct evaluation)
__device__ __forceinline__ SinCos rotate_sincos(const SinCos &a, double d)
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    const SinCos k = kernel_sincos(d);
    SinCos o;
    o.s = fma(a.c, k.s, a.s * k.c);
    o.c = fma(-a.s, k.s, a.c * k.c);
    return o;
}

__device__ __noinline__ SinCos sincos_any(double x) // some lane is outside the fast range
{
    const SinCos slow = ocml_sincos(x);
    const SinCos fast = lean_sincos(fabs(x) < 1.0e5 ? x : 0.0);
    return fabs(x) < 1.0e5 ? fast : slow;
}
__device__ __forceinline__ SinCos m_sincos(double x)
{
    // wave-uniform test: the straight-line lean path unless some lane holds a huge / non-finite angle
    if (__builtin_expect(__ballot(!(fabs(x) < 1.0e5)) == 0ull, 1)) return lean_sincos(x);
    return sincos_any(x);
}
__device__ __forceinline__ double m_sin(double x) { return m_sincos(x).s; }

// The polynomial half of atan_ratio: atan(r) + (off_hi + off_lo) for a reduced |r| <= tan(pi/8)
__device__ __forceinline__ double atan_reduced(double r, double off_hi, double off_lo)
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    const double z = r * r, w = z * z;
    double s1 = fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02);
    s1 = fma(w, s1, 6.66107313738753120669e-02);
    s1 = fma(w, s1, 9.09088713343650656196e-02);
    s1 = fma(w, s1, 1.42857142725034663711e-01);
    s1 = z * fma(w, s1, 3.33333333333329318027e-01);
    double s2 = fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02);
    s2 = fma(w, s2, -7.69187620504482999495e-02);
    s2 = fma(w, s2, -1.11111104054623557880e-01);
    s2 = w * fma(w, s2, -1.99999999998764832476e-01);
    return off_hi - ((r * (s1 + s2) - off_lo) - r);
}

// atan of the ratio num/den of two non-negative numbers (not both zero), result in [0, pi/2]
__device__ __forceinline__ double atan_ratio(double ay, double ax)
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    // three intervals: r = ay/ax, (ay-ax)/(ay+ax) or -ax/ay, |r| <= tan(pi/8)
    const bool lo = ay <= 0.41421356237309503 * ax;
    const bool hi = ay > 2.4142135623730951 * ax;
    const double num = lo ? ay : (hi ? -ax : ay - ax);
    const double den = lo ? ax : (hi ? ay : ay + ax);
    const double r = num / den;
    const double off_hi = lo ? 0.0 : (hi ? 1.57079632679489655800e+00 : 7.85398163397448278999e-01);
    const double off_lo = lo ? 0.0 : (hi ? 6.12323399573676603587e-17 : 3.06161699786838301793e-17);
    return atan_reduced(r, off_hi, off_lo);
}

MPC_ATAN_FN double m_atan(double x) // leaf
{
    const double a = fabs(x);
#ifndef MPC_ATAN_ALWAYS_DIVIDES
    // Every lane inside atan_ratio(a, 1)'s first interval (the Pacejka axle terms bf alpha_f, br alpha_r always are:
    // |alpha| < 1.5 rad): its quotient is a / 1.0 = a exactly, so the division -- eleven instructions that hold a SIMD for
    // ~60 cycles, as long as fourteen fma (tools/micro/fp64_chain.hip) -- is left out.  Same bits; wave-uniform, and a
    // lane's result does not depend on the path its wave takes.
    if (__builtin_expect(__ballot(!(a <= 0.41421356237309503)) == 0ull, 1)) return copysign(atan_reduced(a, 0.0, 0.0), x);
#endif
    const double t = a < 1.0e300 ? atan_ratio(a, 1.0) : 1.57079632679489655800e+00;
    return x != x ? x : copysign(t, x);
}

MPC_ATAN_FN double lean_atan2(double y, double x) // finite, not both zero; a leaf
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    double t = atan_ratio(fabs(y), fabs(x));
    if (signbit(x)) t = 3.14159265358979311600e+00 - (t - 1.22464679914735317720e-16);
    return copysign(t, y);
}
__device__ __forceinline__ double m_atan2(double y, double x)
{
    const double ay = fabs(y), ax = fabs(x);
    const bool ok = ay < 1.0e300 && ax < 1.0e300 && (ay != 0.0 || ax != 0.0);
    if (__builtin_expect(__ballot(!ok) == 0ull, 1)) return lean_atan2(y, x);
    const double slow = ocml_atan2(y, x); // zeros, infinities, NaN in some lane
    const double fast = lean_atan2(ok ? y : 1.0, ok ? x : 1.0);
    return ok ? fast : slow;
}

// ---------------------------------------------------------------------------------- inputs
// Everything that depends on the stage input u only is computed once per stage: u is held
// constant over the nfe*4 RHS evaluations of a stage (and over their adjoints).
template <int MODEL> struct StageInput;

template <> struct StageInput<KIN> {
    double ad;      // accel * d
    double beta;    // slip angle atan2(lf tan(delta), lf + lr)   (dynamics.py:166)
    double sb_lr;   // sin(beta) / lr
    double cb_lr;   // cos(beta) / lr
    double dbeta;   // d beta / d delta
    double mk0, mk1; // clip masks (1 inside the box)
};
template <> struct StageInput<PAC> {
    double d, dl, sd, cd; // drive, steering, sin/cos(steering)
    double mk0, mk1;
};

MPC_DEV void clip_input(const DevCfg &c, double &d, double &dl, double &mk0, double &mk1)
{
    mk0 = 1.0; mk1 = 1.0;
    if (c.clip_inputs) { // dynamics.py:57-65
        if (d > c.max_drive) { d = c.max_drive; mk0 = 0.0; }
        else if (d < -c.max_drive) { d = -c.max_drive; mk0 = 0.0; }
        if (dl > c.max_steer) { dl = c.max_steer; mk1 = 0.0; }
        else if (dl < -c.max_steer) { dl = -c.max_steer; mk1 = 0.0; }
    }
}

MPC_DEV void prep_input(const DevCfg &c, double d, double dl, StageInput<KIN> &s)
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    clip_input(c, d, dl, s.mk0, s.mk1);
    const double L = c.lf + c.lr;
    // tan(delta) from sin/cos, and sin/cos(beta) straight from the triangle (t, L) instead of through
    // the angle.  Three tiers, each guarded by a wave-uniform test so that the common case is one
    // straight-line block: steering angles inside the box (|delta| <= 0.32 by default) take the reduced
    // minimax kernels as they are; anything finite takes the same kernels behind the Cody-Waite
    // reduction -- line-search trial points leave the box, and a wave that paid the library route for
    // one such lane (~1000 instructions per stage) held back its whole launch (profiles/r02a trace:
    // K1a 82 us on first-round data, 140 - 247 us in mid-solve); only non-finite or huge angles go to
    // the library.  What a lane computes does not depend on its neighbours.
    const bool ok = fabs(dl) <= 0.75 && L > 0.0;
    SinCos sd = kernel_sincos(ok ? dl : 0.0);
    bool mid = false;
    if (__builtin_expect(__ballot(!ok) != 0ull, 0)) {
        mid = !ok && fabs(dl) < 1.0e5 && L > 0.0;
        const SinCos sr = lean_sincos(mid ? dl : 0.0);
        sd.s = mid ? sr.s : sd.s; sd.c = mid ? sr.c : sd.c;
    }
    const bool fast = ok || mid;
    double td = sd.s / sd.c;
    double t = c.lf * td;
    const double hyp2 = t * t + L * L;
    const double rh = 1.0 / sqrt(hyp2);
    double sb = t * rh, cb = L * rh;
    double beta;
    if (__builtin_expect(__ballot(!fast) == 0ull, 1)) beta = lean_atan2(t, L);
    else {
        const double td_s = m_tan(dl), t_s = c.lf * td_s;
        const double beta_s = m_atan2(t_s, L);
        const SinCos scb = m_sincos(beta_s);
        const double beta_f = lean_atan2(fast ? t : 0.0, fast ? L : 1.0);
        beta = fast ? beta_f : beta_s;
        td = fast ? td : td_s; t = fast ? t : t_s;
        sb = fast ? sb : scb.s; cb = fast ? cb : scb.c;
    }
    s.beta = beta;
    s.sb_lr = sb / c.lr;
    s.cb_lr = cb / c.lr;
    s.dbeta = (L / (t * t + L * L)) * c.lf * (1.0 + td * td);
    s.ad = c.accel * d;
}

MPC_DEV void prep_input(const DevCfg &c, double d, double dl, StageInput<PAC> &s)
{
    clip_input(c, d, dl, s.mk0, s.mk1);
    s.d = d; s.dl = dl;
    const SinCos sc = m_sincos(dl);
    s.sd = sc.s; s.cd = sc.c;
}

// ---------------------------------------------------------------------------------- RHS
// Lin<MODEL> keeps the local partials of f at one point so that the adjoint never repeats a
// transcendental already evaluated for the forward value.
template <int MODEL> struct Lin;
template <> struct Lin<KIN> { double s, co, v; };
template <> struct Lin<PAC> {
    double sp, cp, f0, f1;     // sin/cos(phi), xdot, ydot
    double vx, vy, om;
    double ffy, Df, Dr;        // front lateral force, d ffy / d alpha_f, d fry / d alpha_r
    double vx_r1, a1_r1, vx_r2, a2_r2; // atan2 partials
};

// kinematic bicycle, dynamics.py:166-172.  x = [x, y, phi, v]
template <bool LIN>
MPC_DEV void rhs(const DevCfg &c, const StageInput<KIN> &u, const double (&x)[4], double (&k)[4],
                 Lin<KIN> &lin)
{
    const SinCos sc = m_sincos(x[2] + u.beta);
    const double s = sc.s, co = sc.c;
    const double v = x[3];
    k[0] = v * co;
    k[1] = v * s;
    k[2] = v * u.sb_lr;
    k[3] = u.ad - c.friction * v;
    if (LIN) { lin.s = s; lin.co = co; lin.v = v; }
}

// The same with sin/cos(phi + beta) supplied by the caller (step_trig below).
template <bool LIN>
MPC_DEV void rhs_sc(const DevCfg &c, const StageInput<KIN> &u, const double (&x)[4], const SinCos &sc,
                    double (&k)[4], Lin<KIN> &lin)
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    const double s = sc.s, co = sc.c;
    const double v = x[3];
    k[0] = v * co;
    k[1] = v * s;
    k[2] = v * u.sb_lr;
    k[3] = u.ad - c.friction * v;
    if (LIN) { lin.s = s; lin.co = co; lin.v = v; }
}

// The heading and speed of the kinematic model do not depend on the position, so the four angles
// phi_i + beta an RK4 step evaluates are known before any sin/cos is: theta_1 = phi + beta and
// theta_i = theta_1 + d_i with the small increments d_2 = h/2 k1_phi, d_3 = h/2 k2_phi, d_4 = h k3_phi.
// One reduced sincos + three rotations replace four dependent full evaluations (a third of the
// instructions, and the four are independent of each other: issue-level parallelism for the
// thread-per-agent rollout).  Falls back to the direct evaluations when an increment is large.
struct StepTrig { SinCos sc[4]; };
MPC_DEV void step_trig(const DevCfg &c, const StageInput<KIN> &u, const double (&x)[4], StepTrig &tr)
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    const double h = c.h, hh = 0.5 * h;
    const double v1 = x[3];
    const double v2 = v1 + hh * (u.ad - c.friction * v1);
    const double v3 = v1 + hh * (u.ad - c.friction * v2);
    const double d2 = hh * (v1 * u.sb_lr), d3 = hh * (v2 * u.sb_lr), d4 = h * (v3 * u.sb_lr);
    const double th = x[2] + u.beta;
    const bool ok = fabs(d2) <= 0.75 && fabs(d3) <= 0.75 && fabs(d4) <= 0.75 && fabs(th) < 1.0e5;
    if (__builtin_expect(__ballot(!ok) == 0ull, 1)) {
        tr.sc[0] = lean_sincos(th);
        tr.sc[1] = rotate_sincos(tr.sc[0], d2);
        tr.sc[2] = rotate_sincos(tr.sc[0], d3);
        tr.sc[3] = rotate_sincos(tr.sc[0], d4);
    } else {
        // Some lane of the wave is outside the fast range.  What a lane computes must not depend on
        // the company it keeps (the work lists are not ordered the same from run to run), so the
        // lanes that are inside still take the rotations; the others get the angles exactly as the
        // plain RK4 forms them.
        const double ths = ok ? th : 0.0;
        StepTrig f;
        f.sc[0] = lean_sincos(ths);
        f.sc[1] = rotate_sincos(f.sc[0], ok ? d2 : 0.0);
        f.sc[2] = rotate_sincos(f.sc[0], ok ? d3 : 0.0);
        f.sc[3] = rotate_sincos(f.sc[0], ok ? d4 : 0.0);
        const SinCos g0 = m_sincos(th), g1 = m_sincos((x[2] + d2) + u.beta);
        const SinCos g2 = m_sincos((x[2] + d3) + u.beta), g3 = m_sincos((x[2] + d4) + u.beta);
        tr.sc[0] = ok ? f.sc[0] : g0; tr.sc[1] = ok ? f.sc[1] : g1;
        tr.sc[2] = ok ? f.sc[2] : g2; tr.sc[3] = ok ? f.sc[3] : g3;
    }
}

// yb[0..1] = adjoint of (phi, v) ; ub += adjoint of (d, delta)
MPC_DEV void vjp(const DevCfg &c, const StageInput<KIN> &u, const Lin<KIN> &l, const double (&w)[4],
                 double (&yb)[4], double (&ub)[2])
{
    const double A = l.v * (w[1] * l.co - w[0] * l.s);
    yb[0] = 0.0; yb[1] = 0.0;
    yb[2] = A;
    yb[3] = w[0] * l.co + w[1] * l.s + w[2] * u.sb_lr - w[3] * c.friction;
    const double beta_b = A + w[2] * l.v * u.cb_lr;
    ub[0] += w[3] * c.accel * u.mk0;
    ub[1] += beta_b * u.dbeta * u.mk1;
}

MPC_DEV double sign_of(double v) { return (v > 0.0 ? 1.0 : 0.0) - (v < 0.0 ? 1.0 : 0.0); }

// car_dynamics.py:121-129 from the heading's and the two axles' sines/cosines: drivetrain force, lateral
// forces, the six derivatives.  Fixed roundings: the thread-per-request RHS and the four-lane RHS
// (rhs_quad) must agree bit for bit.
MPC_DEV void pac_assemble(const DevCfg &c, const StageInput<PAC> &u, double vx, double vy, double om, double sp,
                          double cp, double stf, double str, double (&k)[6])
{
#pragma clang fp contract(off)
    const double frx = (c.cm1 - c.cm2 * vx) * u.d - c.cr0 * sign_of(vx) - c.cr2 * vx * vx;
    const double ffy = c.df * stf;
    const double fry = c.dr * str;
    k[0] = vx * cp - vy * sp;
    k[1] = vx * sp + vy * cp;
    k[2] = om;
    k[3] = (frx - ffy * u.sd + c.mass * vy * om) * c.inv_mass;
    k[4] = (fry + ffy * u.cd - c.mass * vx * om) * c.inv_mass;
    k[5] = (ffy * c.lf * u.cd - fry * c.lr) * c.inv_iz;
}

// Pacejka bicycle, car_dynamics.py:115-129.  x = [x, y, phi, vx, vy, omega]
template <bool LIN>
MPC_DEV void rhs(const DevCfg &c, const StageInput<PAC> &u, const double (&x)[6], double (&k)[6],
                 Lin<PAC> &lin)
{
    const double vx = x[3], vy = x[4], om = x[5];
    const double a1 = om * c.lf + vy;
    const double a2 = om * c.lr - vy;
    // One wave-uniform range test for the whole evaluation instead of one per transcendental: the
    // fast path is then a single straight-line block in which the heading, front-axle and rear-axle
    // chains overlap (the thread-per-agent rollout is bound by the latency of exactly these chains).
    // Per lane the result does not depend on which path the wave takes.
    const double fa1 = fabs(a1), fa2 = fabs(a2), fvx = fabs(vx);
    const bool ok = fabs(x[2]) < 1.0e5 && fa1 < 1.0e300 && fa2 < 1.0e300 && fvx < 1.0e300 &&
                    (fvx != 0.0 || (fa1 != 0.0 && fa2 != 0.0)) && fabs(u.dl) < 1.0e5 &&
                    fabs(c.cf) < 6.0e4 && fabs(c.cr) < 6.0e4; // |cf atan(.)| stays inside lean_sincos' range
    double sp, cp, af, ar, stf, ctf, str, ctr;
    if (__builtin_expect(__ballot(!ok) == 0ull, 1)) {
        const SinCos scp = lean_sincos(x[2]);
        sp = scp.s; cp = scp.c;
        af = u.dl - lean_atan2(a1, vx);
        ar = lean_atan2(a2, vx);
        const SinCos a = lean_sincos(c.cf * m_atan(c.bf * af)), b = lean_sincos(c.cr * m_atan(c.br * ar));
        stf = a.s; ctf = a.c; str = b.s; ctr = b.c;
    } else {
        const SinCos scp = m_sincos(x[2]);
        sp = scp.s; cp = scp.c;
        af = u.dl - m_atan2(a1, vx);
        ar = m_atan2(a2, vx);
        const SinCos a = m_sincos(c.cf * m_atan(c.bf * af)), b = m_sincos(c.cr * m_atan(c.br * ar));
        stf = a.s; ctf = a.c; str = b.s; ctr = b.c;
    }
    const double ffy = c.df * stf;
    pac_assemble(c, u, vx, vy, om, sp, cp, stf, str, k);
    if (LIN) {
        lin.sp = sp; lin.cp = cp; lin.f0 = k[0]; lin.f1 = k[1];
        lin.vx = vx; lin.vy = vy; lin.om = om;
        lin.ffy = ffy;
        const double baf = c.bf * af, bar = c.br * ar;
        lin.Df = c.df * ctf * c.cf * c.bf / (1.0 + baf * baf);
        lin.Dr = c.dr * ctr * c.cr * c.br / (1.0 + bar * bar);
        const double ir1 = 1.0 / (a1 * a1 + vx * vx), ir2 = 1.0 / (a2 * a2 + vx * vx);
        lin.vx_r1 = vx * ir1; lin.a1_r1 = a1 * ir1;
        lin.vx_r2 = vx * ir2; lin.a2_r2 = a2 * ir2;
    }
}

MPC_DEV void vjp(const DevCfg &c, const StageInput<PAC> &u, const Lin<PAC> &l, const double (&w)[6],
                 double (&yb)[6], double (&ub)[2])
{
    // sign(vx) is a constant for the adjoint, exactly as CasADi's AD treats it
    const double w3m = w[3] * c.inv_mass, w4m = w[4] * c.inv_mass, w5z = w[5] * c.inv_iz;
    const double ffy_b = -w3m * u.sd + w4m * u.cd + w5z * c.lf * u.cd;
    const double fry_b = w4m - w5z * c.lr;
    const double af_b = ffy_b * l.Df;
    const double ar_b = fry_b * l.Dr;
    const double a1_b = -af_b * l.vx_r1;
    const double a2_b = ar_b * l.vx_r2;
    yb[0] = 0.0; yb[1] = 0.0;
    yb[2] = w[1] * l.f0 - w[0] * l.f1;
    yb[3] = w[0] * l.cp + w[1] * l.sp - w[4] * l.om + w3m * (-c.cm2 * u.d - 2.0 * c.cr2 * l.vx) +
            af_b * l.a1_r1 - ar_b * l.a2_r2;
    yb[4] = -w[0] * l.sp + w[1] * l.cp + w[3] * l.om + a1_b - a2_b;
    yb[5] = w[2] + w[3] * l.vy - w[4] * l.vx + a1_b * c.lf + a2_b * c.lr;
    const double dl_b = af_b - l.ffy * (w3m * u.cd + w4m * u.sd + w5z * c.lf * u.sd);
    ub[0] += w3m * (c.cm1 - c.cm2 * l.vx) * u.mk0;
    ub[1] += dl_b * u.mk1;
}

// The Pacejka RHS by FOUR lanes (a DPP quad) that all hold the same state and input.  Its cost is
// three chains of transcendentals -- the heading's sin/cos, and per axle atan2 -> atan -> sin/cos -- and
// a lone wave issues them one after the other (340 dependent-ish fp64 instructions per evaluation, 16 N
// evaluations per rollout: the latency that bounds a Pacejka solve's tail).  Here lane role 1 walks the
// front axle's chain, role 2 (and 3) the rear axle's, role 0 the heading's -- ONE instruction stream:
// the roles differ only in the operands they select -- the three results are exchanged by quad
// broadcasts, and every lane assembles the derivatives, so all four keep identical copies of the
// state.  Same operations on the same values as rhs<false>: bit-identical.
template <int CTRL> MPC_DEV double quad_bcast(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
MPC_DEV void rhs_quad(const DevCfg &c, const StageInput<PAC> &u, const double (&x)[6], double (&k)[6], int role)
{
    const double vx = x[3], vy = x[4], om = x[5];
    const double a1 = om * c.lf + vy;
    const double a2 = om * c.lr - vy;
    const double fa1 = fabs(a1), fa2 = fabs(a2), fvx = fabs(vx);
    const bool ok = fabs(x[2]) < 1.0e5 && fa1 < 1.0e300 && fa2 < 1.0e300 && fvx < 1.0e300 &&
                    (fvx != 0.0 || (fa1 != 0.0 && fa2 != 0.0)) && fabs(u.dl) < 1.0e5 &&
                    fabs(c.cf) < 6.0e4 && fabs(c.cr) < 6.0e4;
    if (__builtin_expect(__ballot(!ok) != 0ull, 0)) {      // some lane is out of the fast ranges: every
        Lin<PAC> dummy;                                     // lane evaluates the whole RHS by itself
        rhs<false>(c, u, x, k, dummy);
        return;
    }
    const bool front = role == 1;
    const double t = lean_atan2(front ? a1 : a2, vx);
    const double al = front ? u.dl - t : t;                 // alpha_f = delta - atan2(a1, vx); alpha_r = atan2(a2, vx)
    const double axle = (front ? c.cf : c.cr) * m_atan((front ? c.bf : c.br) * al);
    const SinCos sc = lean_sincos(role == 0 ? x[2] : axle);
    const double sp = quad_bcast<0x00>(sc.s), cp = quad_bcast<0x00>(sc.c);
    const double stf = quad_bcast<0x55>(sc.s), str = quad_bcast<0xAA>(sc.s);
    pac_assemble(c, u, vx, vy, om, sp, cp, stf, str, k);
}

// ---------------------------------------------------------------------------------- RK4
template <int MODEL> struct ModelDim { static constexpr int NX = MODEL == PAC ? 6 : 4; };

// the i-th RHS evaluation of an RK4 step: the kinematic model takes its sin/cos from step_trig
template <bool LIN, int I>
MPC_DEV void rk_rhs(const DevCfg &c, const StageInput<KIN> &u, const StepTrig &tr, const double (&x)[4],
                    double (&k)[4], Lin<KIN> &lin)
{
    rhs_sc<LIN>(c, u, x, tr.sc[I], k, lin);
}
template <bool LIN, int I>
MPC_DEV void rk_rhs(const DevCfg &c, const StageInput<PAC> &u, const StepTrig &, const double (&x)[6],
                    double (&k)[6], Lin<PAC> &lin)
{
    rhs<LIN>(c, u, x, k, lin);
}
MPC_DEV void step_trig(const DevCfg &, const StageInput<PAC> &, const double (&)[6], StepTrig &) {}

// one classical RK4 step (car_dynamics.py:136-145, h = Ts / nfe, input held) of the Pacejka model around
// an RHS evaluator f(x, k) -- the thread-per-request one or the four-lane one.  Fixed roundings: the
// two must produce the same bits.
template <class F>
MPC_DEV void rk4_step_pac(const DevCfg &c, double (&x)[6], F f)
{
#pragma clang fp contract(off)
    const double h = c.h, hh = 0.5 * h, h6 = h / 6.0;
    double k1[6], k2[6], k3[6], k4[6], t[6];
    f(x, k1);
#pragma unroll
    for (int i = 0; i < 6; i++) t[i] = fma(hh, k1[i], x[i]);
    f(t, k2);
#pragma unroll
    for (int i = 0; i < 6; i++) t[i] = fma(hh, k2[i], x[i]);
    f(t, k3);
#pragma unroll
    for (int i = 0; i < 6; i++) t[i] = fma(h, k3[i], x[i]);
    f(t, k4);
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = fma(h6, ((k1[i] + 2.0 * k2[i]) + 2.0 * k3[i]) + k4[i], x[i]);
}

// A trial point of the line search can make the Pacejka model blow up inside the horizon (RK4 at this step size is
// only stable near the road): heading, velocities and yaw rate all NaN.  Every derivative of such a state is NaN
// (each of the six reads at least one of the four), so the state stays where it is for the rest of the horizon --
// but every one of its 16 evaluations per stage would drag its whole wave through the library route of the range
// tests in rhs / rhs_quad (mid-solve: 3 % of the rollout waves, five times as long as the others, set the length of
// every launch).  A stage that STARTS there is therefore not evaluated: the lanes walk a harmless state instead and
// get their NaNs back afterwards.
MPC_DEV bool pac_state_is_lost(const double (&x)[6]) { return x[2] != x[2] && x[3] != x[3] && x[4] != x[4] && x[5] != x[5]; }
// ... and so is a stage whose steering input is NaN or infinite, or whose drive is NaN (the trial point itself was
// formed from a NaN direction): sin / cos of the steering, or the drivetrain force, put a NaN into all three
// accelerations at once, the heading follows the yaw rate within the RK4 step, the position the velocities
MPC_DEV bool pac_stage_is_lost(const StageInput<PAC> &u, const double (&x)[6])
{
    return pac_state_is_lost(x) || !(fabs(u.dl) < INFINITY) || u.d != u.d;
}
MPC_DEV void pac_park(StageInput<PAC> &u, double (&x)[6])
{
    x[2] = 0.0; x[3] = 1.0; x[4] = 0.0; x[5] = 0.0;
    u.d = 0.0; u.dl = 0.0; u.sd = 0.0; u.cd = 1.0;
}
MPC_DEV void pac_lose(double (&x)[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = __builtin_nan("");
}
// one stage x <- f_d(x, u) : nfe RK4 steps
MPC_DEV void stage_forward_steps(const DevCfg &c, const StageInput<PAC> &u_in, double (&x)[6])
{
    StageInput<PAC> u = u_in;
    const bool lost = pac_stage_is_lost(u, x);
    if (lost) pac_park(u, x);
    for (int s = 0; s < c.nfe; s++)
        rk4_step_pac(c, x, [&](const double (&y)[6], double (&k)[6]) { Lin<PAC> dummy; rhs<false>(c, u, y, k, dummy); });
    if (lost) pac_lose(x);
}
// the same by a quad of lanes holding identical (x, u); role = lane & 3 (see rhs_quad)
MPC_DEV void stage_forward_quad(const DevCfg &c, const StageInput<PAC> &u_in, double (&x)[6], int role)
{
    StageInput<PAC> u = u_in;
    const bool lost = pac_stage_is_lost(u, x);
    if (lost) pac_park(u, x);
    for (int s = 0; s < c.nfe; s++)
        rk4_step_pac(c, x, [&](const double (&y)[6], double (&k)[6]) { rhs_quad(c, u, y, k, role); });
    if (lost) pac_lose(x);
}
MPC_DEV void stage_forward_steps(const DevCfg &c, const StageInput<KIN> &u, double (&x)[4])
{
#pragma clang fp contract(off)   // fixed roundings (see kin_rk): this path is reachable from two kernels
    const double h = c.h;
    for (int s = 0; s < c.nfe; s++) {
        double k1[4], k2[4], k3[4], k4[4], t[4];
        Lin<KIN> dummy;
        StepTrig tr;
        step_trig(c, u, x, tr);
        rhs_sc<false>(c, u, x, tr.sc[0], k1, dummy);
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = x[i] + 0.5 * h * k1[i];
        rhs_sc<false>(c, u, t, tr.sc[1], k2, dummy);
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = x[i] + 0.5 * h * k2[i];
        rhs_sc<false>(c, u, t, tr.sc[2], k3, dummy);
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = x[i] + h * k3[i];
        rhs_sc<false>(c, u, t, tr.sc[3], k4, dummy);
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = x[i] + (h / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
}

// The kinematic stage with nfe = 4 as ONE straight-line block.  Heading and speed obey a linear
// recursion that needs no sin/cos, so the four angles of an RK4 step are known before any of them is
// evaluated: its sin/cos pairs (1 reduction + 3 rotations) are independent of each other and of the
// position sums.  Same operations as rk4_step<KIN> four times -- only the schedule changes: the
// thread-per-agent rollout is bound by the latency of a dependent chain, and this cuts the chain of
// an RK4 step from four sin/cos to one.
// One RK4 step of the kinematic model in three pieces with fixed roundings (no contraction), so that
// the thread-per-agent rollout and the wave-per-agent rollout (rollout_wide_kernel) produce the same
// bits: the speed/heading stage values, the step to the next (heading, speed), and the position
// increment (1 reduced sin/cos + 3 rotations).
struct KinRK { double v1, v2, v3, v4, kp1, kp2, kp3, kp4, kv1, kv2, kv3, kv4; };
MPC_DEV void kin_rk(const DevCfg &c, const StageInput<KIN> &u, double v, KinRK &k)
{
#pragma clang fp contract(off)
    // explicit fma: nine dependent operations from v to the next v (this recursion is the serial
    // part of the wave-per-request rollout)
    const double h = c.h, hh = 0.5 * h, nf = -c.friction;
    k.v1 = v;                    k.kv1 = fma(nf, k.v1, u.ad); k.kp1 = k.v1 * u.sb_lr;
    k.v2 = fma(hh, k.kv1, v);    k.kv2 = fma(nf, k.v2, u.ad); k.kp2 = k.v2 * u.sb_lr;
    k.v3 = fma(hh, k.kv2, v);    k.kv3 = fma(nf, k.v3, u.ad); k.kp3 = k.v3 * u.sb_lr;
    k.v4 = fma(h, k.kv3, v);     k.kv4 = fma(nf, k.v4, u.ad); k.kp4 = k.v4 * u.sb_lr;
}
MPC_DEV void kin_next(const DevCfg &c, const KinRK &k, double &ph, double &v)
{
#pragma clang fp contract(off)
    const double h6 = c.h / 6.0;
    ph = fma(h6, fma(2.0, k.kp3, fma(2.0, k.kp2, k.kp1)) + k.kp4, ph);
    v = fma(h6, fma(2.0, k.kv3, fma(2.0, k.kv2, k.kv1)) + k.kv4, v);
}
MPC_DEV void kin_increment(const DevCfg &c, const StageInput<KIN> &u, double ph, const KinRK &k, bool ok,
                           double &dx, double &dy)
{
#pragma clang fp contract(off)
    const double h = c.h, hh = 0.5 * h, h6 = h / 6.0;
    const double th = ph + u.beta;
    const SinCos a0 = lean_sincos(ok ? th : 0.0);
    const SinCos a1 = rotate_sincos(a0, ok ? hh * k.kp1 : 0.0);
    const SinCos a2 = rotate_sincos(a0, ok ? hh * k.kp2 : 0.0);
    const SinCos a3 = rotate_sincos(a0, ok ? h * k.kp3 : 0.0);
    const double k1x = k.v1 * a0.c, k2x = k.v2 * a1.c, k3x = k.v3 * a2.c, k4x = k.v4 * a3.c;
    const double k1y = k.v1 * a0.s, k2y = k.v2 * a1.s, k3y = k.v3 * a2.s, k4y = k.v4 * a3.s;
    dx = h6 * (fma(2.0, k3x, fma(2.0, k2x, k1x)) + k4x);
    dy = h6 * (fma(2.0, k3y, fma(2.0, k2y, k1y)) + k4y);
}

// The same three pieces with the roundings of stage_forward_steps<KIN> (products and sums kept apart, the
// sin/cos pairs of step_trig), which is what a stage OUTSIDE the fast range (kin4_in_range) is computed by:
// the wave-per-request rollout serves such stages too, bit for bit, instead of handing the whole request to
// one lane.  kin_gen_rk: the speed / heading stage values of one RK4 step; kin_gen_next: the step to the next
// (heading, speed); kin_gen_increment: the position increment.
MPC_DEV void kin_gen_rk(const DevCfg &c, const StageInput<KIN> &u, double v, KinRK &k)
{
#pragma clang fp contract(off)
    const double h = c.h, hh = 0.5 * h;
    k.v1 = v;               k.kv1 = u.ad - c.friction * k.v1; k.kp1 = k.v1 * u.sb_lr;
    k.v2 = v + hh * k.kv1;  k.kv2 = u.ad - c.friction * k.v2; k.kp2 = k.v2 * u.sb_lr;
    k.v3 = v + hh * k.kv2;  k.kv3 = u.ad - c.friction * k.v3; k.kp3 = k.v3 * u.sb_lr;
    k.v4 = v + h * k.kv3;   k.kv4 = u.ad - c.friction * k.v4; k.kp4 = k.v4 * u.sb_lr;
}
MPC_DEV void kin_gen_next(const DevCfg &c, const KinRK &k, double &ph, double &v)
{
#pragma clang fp contract(off)
    const double h = c.h;
    ph = ph + (h / 6.0) * (k.kp1 + 2.0 * k.kp2 + 2.0 * k.kp3 + k.kp4);
    v = v + (h / 6.0) * (k.kv1 + 2.0 * k.kv2 + 2.0 * k.kv3 + k.kv4);
}
MPC_DEV void kin_gen_increment(const DevCfg &c, const StageInput<KIN> &u, double ph, double v, const KinRK &k,
                               double &dx, double &dy)
{
#pragma clang fp contract(off)
    const double h = c.h;
    const double x[4] = {0.0, 0.0, ph, v};
    StepTrig tr;
    step_trig(c, u, x, tr);
    const double k1x = k.v1 * tr.sc[0].c, k2x = k.v2 * tr.sc[1].c, k3x = k.v3 * tr.sc[2].c, k4x = k.v4 * tr.sc[3].c;
    const double k1y = k.v1 * tr.sc[0].s, k2y = k.v2 * tr.sc[1].s, k3y = k.v3 * tr.sc[2].s, k4y = k.v4 * tr.sc[3].s;
    dx = (h / 6.0) * (k1x + 2.0 * k2x + 2.0 * k3x + k4x);
    dy = (h / 6.0) * (k1y + 2.0 * k2y + 2.0 * k3y + k4y);
}

MPC_DEV void stage_forward_kin4(const DevCfg &c, const StageInput<KIN> &u, double (&x)[4], bool ok_in)
{
#pragma clang fp contract(off)
    double ph = x[2], v = x[3], px = x[0], py = x[1];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        KinRK k;
        kin_rk(c, u, v, k);
        double dx, dy;
        kin_increment(c, u, ph, k, ok_in, dx, dy);
        px = px + dx;
        py = py + dy;
        kin_next(c, k, ph, v);
        // four independent sin/cos pairs in flight are enough to hide the FP64 latency; letting the
        // scheduler interleave all sixteen costs more registers than the wave has
        __builtin_amdgcn_sched_barrier(0);
    }
    x[0] = px; x[1] = py; x[2] = ph; x[3] = v;
}

// |increment| <= 0.7 (the kernels hold to pi/4) and |angle| < 1e5 for every RK4 step of the stage (what stage_forward_kin4 needs)
MPC_DEV bool kin4_in_range(const DevCfg &c, const StageInput<KIN> &u, const double (&x)[4])
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    // the speed recursion is linear and contractive-ish: bound the increments through max |v| over the
    // stage, |v_i| <= |v| + Ts (|ad| + friction |v|) (1 + ...) -- a cheap sufficient test
    const double Ts = 4.0 * c.h;
    const double vmax = fabs(x[3]) + 2.0 * Ts * (fabs(u.ad) + fabs(c.friction * x[3]));
    const double dmax = c.h * vmax * fabs(u.sb_lr);
    const double amax = fabs(x[2]) + fabs(u.beta) + 4.0 * dmax;
    return dmax <= 0.7 && amax < 1.0e5 && fabs(c.friction) * Ts <= 0.5;
}

template <int MODEL>
MPC_DEV void stage_forward(const DevCfg &c, const StageInput<MODEL> &u,
                           double (&x)[ModelDim<MODEL>::NX])
{
    if constexpr (MODEL == KIN) {
        if (c.nfe == 4) {
            const bool ok = kin4_in_range(c, u, x);
            if (__builtin_expect(__ballot(!ok) == 0ull, 1)) { stage_forward_kin4(c, u, x, true); return; }
            // mixed wave: a lane's result must not depend on its neighbours -- in-range lanes keep the
            // straight-line path, the others take the step-by-step one
            double xa[4] = {x[0], x[1], x[2], x[3]}, xb[4] = {x[0], x[1], x[2], x[3]};
            if (!ok) { xa[2] = 0.0; xa[3] = 0.0; }
            stage_forward_kin4(c, u, xa, ok);
            stage_forward_steps(c, u, xb);
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = ok ? xa[i] : xb[i];
            return;
        }
    }
    stage_forward_steps(c, u, x);
}

// ---------------------------------------------------------------------------------- tangents
// Forward-mode counterpart of vjp(): dk = J_x dy + J_u (dd, ddl) at a point whose partials are in l.
MPC_DEV void jvp(const DevCfg &c, const StageInput<KIN> &u, const Lin<KIN> &l, const double (&dy)[4],
                 double dd, double ddl, double (&dk)[4])
{
    const double dang = dy[2] + u.dbeta * (ddl * u.mk1); // d(phi + beta)
    const double dv = dy[3];
    dk[0] = l.co * dv - l.v * l.s * dang;
    dk[1] = l.s * dv + l.v * l.co * dang;
    dk[2] = u.sb_lr * dv + l.v * u.cb_lr * u.dbeta * (ddl * u.mk1);
    dk[3] = c.accel * (dd * u.mk0) - c.friction * dv;
}

MPC_DEV void jvp(const DevCfg &c, const StageInput<PAC> &u, const Lin<PAC> &l, const double (&dy)[6],
                 double dd_in, double ddl_in, double (&dk)[6])
{
    const double dd = dd_in * u.mk0, ddl = ddl_in * u.mk1;
    const double dphi = dy[2], dvx = dy[3], dvy = dy[4], dom = dy[5];
    const double da1 = dom * c.lf + dvy, da2 = dom * c.lr - dvy;
    const double daf = ddl - (l.vx_r1 * da1 - l.a1_r1 * dvx);
    const double dar = l.vx_r2 * da2 - l.a2_r2 * dvx;
    const double dffy = l.Df * daf, dfry = l.Dr * dar;
    const double dfrx = (c.cm1 - c.cm2 * l.vx) * dd + (-c.cm2 * u.d - 2.0 * c.cr2 * l.vx) * dvx;
    dk[0] = -l.f1 * dphi + l.cp * dvx - l.sp * dvy;
    dk[1] = l.f0 * dphi + l.sp * dvx + l.cp * dvy;
    dk[2] = dom;
    dk[3] = (dfrx - dffy * u.sd - l.ffy * u.cd * ddl) * c.inv_mass + dvy * l.om + l.vy * dom;
    dk[4] = (dfry + dffy * u.cd - l.ffy * u.sd * ddl) * c.inv_mass - dvx * l.om - l.vx * dom;
    dk[5] = (dffy * c.lf * u.cd - l.ffy * c.lf * u.sd * ddl - dfry * c.lr) * c.inv_iz;
}

// Sensitivities of one stage x+ = f_d(xs, u): T[d][comp] = d x+_comp / d dir_d, directions
// d = 0..NX-3: the non-position states (phi, v | phi, vx, vy, omega); d = NX-2, NX-1: the inputs
// (d, delta).  The position states enter f_d additively, so their columns are the identity.
template <int MODEL>
MPC_DEV void stage_tangents(const DevCfg &c, const StageInput<MODEL> &u,
                            const double (&xs)[ModelDim<MODEL>::NX],
                            double (&T)[ModelDim<MODEL>::NX][ModelDim<MODEL>::NX])
{
    constexpr int NX = ModelDim<MODEL>::NX, NZ = NX - 2;
    const double h = c.h;
    double x[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = xs[i];
#pragma unroll
    for (int d = 0; d < NX; d++) {
#pragma unroll
        for (int i = 0; i < NX; i++) T[d][i] = (d < NZ && i == d + 2) ? 1.0 : 0.0;
    }
    for (int s = 0; s < c.nfe; s++) {
        double k1[NX], k2[NX], k3[NX], k4[NX], t[NX];
        Lin<MODEL> l1, l2, l3, l4;
        StepTrig tr;
        step_trig(c, u, x, tr);
        rk_rhs<true, 0>(c, u, tr, x, k1, l1);
#pragma unroll
        for (int i = 0; i < NX; i++) t[i] = x[i] + 0.5 * h * k1[i];
        rk_rhs<true, 1>(c, u, tr, t, k2, l2);
#pragma unroll
        for (int i = 0; i < NX; i++) t[i] = x[i] + 0.5 * h * k2[i];
        rk_rhs<true, 2>(c, u, tr, t, k3, l3);
#pragma unroll
        for (int i = 0; i < NX; i++) t[i] = x[i] + h * k3[i];
        rk_rhs<true, 3>(c, u, tr, t, k4, l4);
#pragma unroll
        for (int d = 1; d < NX; d++) { // the heading direction (d = 0) is analytic, see below
            const double dd = d == NZ ? 1.0 : 0.0, ddl = d == NZ + 1 ? 1.0 : 0.0;
            double d1[NX], d2[NX], d3[NX], d4[NX], dt[NX];
            jvp(c, u, l1, T[d], dd, ddl, d1);
#pragma unroll
            for (int i = 0; i < NX; i++) dt[i] = T[d][i] + 0.5 * h * d1[i];
            jvp(c, u, l2, dt, dd, ddl, d2);
#pragma unroll
            for (int i = 0; i < NX; i++) dt[i] = T[d][i] + 0.5 * h * d2[i];
            jvp(c, u, l3, dt, dd, ddl, d3);
#pragma unroll
            for (int i = 0; i < NX; i++) dt[i] = T[d][i] + h * d3[i];
            jvp(c, u, l4, dt, dd, ddl, d4);
#pragma unroll
            for (int i = 0; i < NX; i++) T[d][i] += (h / 6.0) * (d1[i] + 2.0 * d2[i] + 2.0 * d3[i] + d4[i]);
        }
#pragma unroll
        for (int i = 0; i < NX; i++) x[i] = x[i] + (h / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    // Heading direction: both models are equivariant under a rotation of the frame -- every angle an
    // RK4 stage evaluates is phi + (something that does not depend on phi), the other states never see
    // phi -- so d(x+ - x, y+ - y)/d phi is the displacement turned by 90 degrees, exactly, for the
    // discrete map too.  One of the NX tangent directions costs nothing.
    T[0][0] = -(x[1] - xs[1]);
    T[0][1] = x[0] - xs[0];
    T[0][2] = 1.0;
#pragma unroll
    for (int i = 3; i < NX; i++) T[0][i] = 0.0;
}

// The kinematic stage with nfe = 4 inside the fast range (kin4_in_range): the sensitivities written out
// for this model instead of four generic Jacobian-vector products per RK4 step.  Speed and heading do not
// see the position, the speed does not see the steering, and the position only collects increments,
// so per RK4 step a direction needs: its speed chain (4 values; none for the steering direction), its
// heading chain (4 values), and the increments d(kx_i, ky_i) = (c_i dv_i - ky_i dtheta_i,
// s_i dv_i + kx_i dtheta_i).  Same stage values (kin_rk) and sin/cos pairs (one reduction + three
// rotations) as the forward rollout.  Far fewer live values than the generic form (no Lin records, no k
// arrays, no structural zeros): K1b's gradient blocks fit three waves per SIMD without spilling.
// xe is the stage's end state AS THE ROLLOUT COMPUTED IT: the heading direction is the displacement
// (xe - xs) turned by 90 degrees (see stage_tangents).  Fixed roundings throughout.
MPC_DEV void stage_tangents_kin4(const DevCfg &c, const StageInput<KIN> &u, const double (&xs)[4],
                                 const double (&xe)[4], double (&T)[4][4])
{
#pragma clang fp contract(off)
    const double h = c.h, hh = 0.5 * h, h6 = h / 6.0, nf = -c.friction;
    const double A = c.accel * u.mk0;           // d vdot / d drive
    const double Cd = u.dbeta * u.mk1;          // d (phi + beta) / d delta
    const double Bd = u.cb_lr * Cd;             // d (phidot / v) / d delta
    double ph = xs[2], v = xs[3];
    double ax = 0.0, ay = 0.0, ap = 0.0, av = 1.0;   // direction: speed
    double bx = 0.0, by = 0.0, bp = 0.0, bv = 0.0;   // direction: drive
    double ex = 0.0, ey = 0.0, ep = 0.0;             // direction: steering (its speed tangent is zero)
#pragma unroll
    for (int s = 0; s < 4; s++) {
        KinRK k;
        kin_rk(c, u, v, k);
        const SinCos a0 = lean_sincos(ph + u.beta);
        const SinCos a1 = rotate_sincos(a0, hh * k.kp1), a2 = rotate_sincos(a0, hh * k.kp2), a3 = rotate_sincos(a0, h * k.kp3);
        const double k1x = k.v1 * a0.c, k2x = k.v2 * a1.c, k3x = k.v3 * a2.c, k4x = k.v4 * a3.c;
        const double k1y = k.v1 * a0.s, k2y = k.v2 * a1.s, k3y = k.v3 * a2.s, k4y = k.v4 * a3.s;
        {   // speed direction
            const double d1 = av, e1 = nf * d1, d2 = fma(hh, e1, av), e2 = nf * d2, d3 = fma(hh, e2, av), e3 = nf * d3,
                         d4 = fma(h, e3, av), e4 = nf * d4;
            const double p1 = d1 * u.sb_lr, p2 = d2 * u.sb_lr, p3 = d3 * u.sb_lr, p4 = d4 * u.sb_lr;
            const double f1 = ap, f2 = fma(hh, p1, ap), f3 = fma(hh, p2, ap), f4 = fma(h, p3, ap);
            const double g1x = fma(-k1y, f1, a0.c * d1), g2x = fma(-k2y, f2, a1.c * d2), g3x = fma(-k3y, f3, a2.c * d3),
                         g4x = fma(-k4y, f4, a3.c * d4);
            const double g1y = fma(k1x, f1, a0.s * d1), g2y = fma(k2x, f2, a1.s * d2), g3y = fma(k3x, f3, a2.s * d3),
                         g4y = fma(k4x, f4, a3.s * d4);
            ax = fma(h6, fma(2.0, g3x, fma(2.0, g2x, g1x)) + g4x, ax);
            ay = fma(h6, fma(2.0, g3y, fma(2.0, g2y, g1y)) + g4y, ay);
            ap = fma(h6, fma(2.0, p3, fma(2.0, p2, p1)) + p4, ap);
            av = fma(h6, fma(2.0, e3, fma(2.0, e2, e1)) + e4, av);
        }
        {   // drive direction
            const double d1 = bv, e1 = fma(nf, d1, A), d2 = fma(hh, e1, bv), e2 = fma(nf, d2, A), d3 = fma(hh, e2, bv),
                         e3 = fma(nf, d3, A), d4 = fma(h, e3, bv), e4 = fma(nf, d4, A);
            const double p1 = d1 * u.sb_lr, p2 = d2 * u.sb_lr, p3 = d3 * u.sb_lr, p4 = d4 * u.sb_lr;
            const double f1 = bp, f2 = fma(hh, p1, bp), f3 = fma(hh, p2, bp), f4 = fma(h, p3, bp);
            const double g1x = fma(-k1y, f1, a0.c * d1), g2x = fma(-k2y, f2, a1.c * d2), g3x = fma(-k3y, f3, a2.c * d3),
                         g4x = fma(-k4y, f4, a3.c * d4);
            const double g1y = fma(k1x, f1, a0.s * d1), g2y = fma(k2x, f2, a1.s * d2), g3y = fma(k3x, f3, a2.s * d3),
                         g4y = fma(k4x, f4, a3.s * d4);
            bx = fma(h6, fma(2.0, g3x, fma(2.0, g2x, g1x)) + g4x, bx);
            by = fma(h6, fma(2.0, g3y, fma(2.0, g2y, g1y)) + g4y, by);
            bp = fma(h6, fma(2.0, p3, fma(2.0, p2, p1)) + p4, bp);
            bv = fma(h6, fma(2.0, e3, fma(2.0, e2, e1)) + e4, bv);
        }
        {   // steering direction
            const double p1 = k.v1 * Bd, p2 = k.v2 * Bd, p3 = k.v3 * Bd, p4 = k.v4 * Bd;
            const double f1 = ep + Cd, f2 = fma(hh, p1, ep) + Cd, f3 = fma(hh, p2, ep) + Cd, f4 = fma(h, p3, ep) + Cd;
            const double sx = fma(2.0, k3y * f3, fma(2.0, k2y * f2, k1y * f1)) + k4y * f4;
            const double sy = fma(2.0, k3x * f3, fma(2.0, k2x * f2, k1x * f1)) + k4x * f4;
            ex = fma(-h6, sx, ex);
            ey = fma(h6, sy, ey);
            ep = fma(h6, fma(2.0, p3, fma(2.0, p2, p1)) + p4, ep);
        }
        kin_next(c, k, ph, v);
        __builtin_amdgcn_sched_barrier(0);      // one RK4 step's values at a time (see stage_forward_kin4)
    }
    T[0][0] = -(xe[1] - xs[1]); T[0][1] = xe[0] - xs[0]; T[0][2] = 1.0; T[0][3] = 0.0;
    T[1][0] = ax; T[1][1] = ay; T[1][2] = ap; T[1][3] = av;
    T[2][0] = bx; T[2][1] = by; T[2][2] = bp; T[2][3] = bv;
    T[3][0] = ex; T[3][1] = ey; T[3][2] = ep; T[3][3] = 0.0;
}

// stage_tangents with the rollout's end state at hand: the kinematic model takes the form above when it
// applies (a lane's result never depends on the other lanes of its wave: a lane outside the fast range
// keeps the generic form while its neighbours keep theirs)
template <int MODEL>
MPC_DEV void stage_tangents_at(const DevCfg &c, const StageInput<MODEL> &u, const double (&xs)[ModelDim<MODEL>::NX],
                               const double (&xe)[ModelDim<MODEL>::NX],
                               double (&T)[ModelDim<MODEL>::NX][ModelDim<MODEL>::NX])
{
    if constexpr (MODEL == KIN) {
        if (c.nfe == 4) {
            const bool ok = kin4_in_range(c, u, xs);
            if (__builtin_expect(__ballot(!ok) == 0ull, 1)) { stage_tangents_kin4(c, u, xs, xe, T); return; }
            double xa[4] = {xs[0], xs[1], ok ? xs[2] : 0.0, ok ? xs[3] : 0.0}, Ta[4][4];
            stage_tangents_kin4(c, u, xa, xe, Ta);
            stage_tangents<MODEL>(c, u, xs, T);
#pragma unroll
            for (int d = 0; d < 4; d++) {
#pragma unroll
                for (int i = 0; i < 4; i++) T[d][i] = ok ? Ta[d][i] : T[d][i];
            }
            return;
        }
    }
    stage_tangents<MODEL>(c, u, xs, T);
}

// ---------------------------------------------------------------------------------- tracking
// car_dynamics.py:174-192: start at point 0, candidates 1..S-2, strict <.  Squared distances are
// compared (sqrt is monotone).  cl is the flat row [x_0..x_{S-1}, y_0..y_{S-1}].
// The squared distance is ONE fixed expression, fma(dx, dx, dy * dy), wherever it is formed: the
// block-pruned search below must reproduce the full scan's argmin bit for bit.
MPC_DEV double dist2(double cx, double cy, double px, double py)
{
#pragma clang fp contract(off)
    const double dx = cx - px, dy = cy - py;
    return fma(dx, dx, dy * dy);
}

MPC_DEV int nearest_index(const DevCfg &c, const double *__restrict__ cl, double px, double py)
{
    const int S = c.S;
    double best = dist2(cl[0], cl[S], px, py);
    int idx = 0;
    int i = 1;
    // four candidates per trip: their distances are independent (ILP for a lone wave); the
    // in-order strict "<" of the reference is kept by resolving ties towards the lower index
    for (; i + 3 < S - 1; i += 4) {
        const double d0 = dist2(cl[i], cl[S + i], px, py), d1 = dist2(cl[i + 1], cl[S + i + 1], px, py);
        const double d2 = dist2(cl[i + 2], cl[S + i + 2], px, py), d3 = dist2(cl[i + 3], cl[S + i + 3], px, py);
        const bool l01 = d1 < d0, l23 = d3 < d2;
        const double m01 = l01 ? d1 : d0, m23 = l23 ? d3 : d2;
        const int i01 = l01 ? i + 1 : i, i23 = l23 ? i + 3 : i + 2;
        const bool lq = m23 < m01;
        const double mq = lq ? m23 : m01;
        const int iq = lq ? i23 : i01;
        const bool lt = mq < best;
        best = lt ? mq : best;
        idx = lt ? iq : idx;
    }
    for (; i < S - 1; i++) {
        const double d2 = dist2(cl[i], cl[S + i], px, py);
        const bool lt = d2 < best;
        best = lt ? d2 : best;
        idx = lt ? i : idx;
    }
    return idx;
}

// the same argmin without looking at every point.  The candidates 0 .. S-2 are cut into
// blocks of NEAR_BLK consecutive points whose bounding boxes [xlo, xhi, ylo, yhi] are computed once per
// centerline row (cl_blocks_kernel).  (1) the first point of every block is a candidate like any
// other: the best of them bounds the minimum from above; (2) a block whose box is farther than that
// bound cannot hold the minimum -- the box distance is formed by the same expression as a point
// distance and rounding is monotone, so "cannot" holds bit for bit; (3) the remaining blocks (one or
// two on a smooth track) are scanned, each lane walking ITS OWN blocks through per-lane loads, so a
// wave of cars spread along the track does not pay for the union of their neighbourhoods.  Ties go
// to the lower index, as the reference's in-order strict "<" does.
constexpr int NEAR_BLK = 8;
MPC_DEV int nearest_index_blocks(const DevCfg &c, const double *__restrict__ cl, const double *__restrict__ bx,
                                 double px, double py)
{
#pragma clang fp contract(off)
    const int S = c.S, nc = S - 1, NB = (nc + NEAR_BLK - 1) / NEAR_BLK;
    double best = dist2(cl[0], cl[S], px, py);
    int idx = 0;
    for (int b = 1; b < NB; b++) {                       // (1) block heads, in index order: strict "<" keeps ties low
        const int i = b * NEAR_BLK;
        const double d = dist2(cl[i], cl[S + i], px, py);
        const bool lt = d < best;
        best = lt ? d : best;
        idx = lt ? i : idx;
    }
    unsigned long long mask = 0ull;
    for (int b = 0; b < NB; b++) {                       // (2) boxes that can still hold the minimum
        const double *q = bx + 4 * b;
        const double ex = fmax(fmax(q[0] - px, px - q[1]), 0.0), ey = fmax(fmax(q[2] - py, py - q[3]), 0.0);
        const double lb = fma(ex, ex, ey * ey);
        mask |= lb <= best ? 1ull << b : 0ull;
    }
    while (__ballot(mask != 0ull) != 0ull) {             // (3) uniform loop, per-lane blocks
        const bool on = mask != 0ull;
        const int b = on ? (int)__builtin_ctzll(mask) : 0;
        mask &= mask - 1ull;
#pragma unroll
        for (int j = 1; j < NEAR_BLK; j++) {             // the head (j = 0) has been looked at
            const int i = b * NEAR_BLK + j;
            const bool valid = on && i < nc;
            const int ii = valid ? i : 0;
            const double d = dist2(cl[ii], cl[S + ii], px, py);
            const bool take = valid && (d < best || (d == best && i < idx));
            best = take ? d : best;
            idx = take ? i : idx;
        }
    }
    return idx;
}

// SURVEY 8f-2, second form: a uniform grid over the neighbourhood of a centerline row.  Every cell holds
// the index range [lo, hi] that is GUARANTEED to contain the full scan's answer for any query point in
// the cell: with U = min_j (largest squared distance from the cell to point j), the nearest point of any
// query in the cell is at most U away, so it is among the points whose smallest squared distance to
// the cell is <= U (cl_grid_cells_kernel adds a relative 1e-9 and grows the cell by 1e-6 of its size:
// far more than the roundings of dist2 and of the cell lookup).  The first minimum of dist2 over
// [lo, hi] is then the first minimum over all candidates -- the same index as nearest_index, bit for
// bit, ties included -- and a lane looks at ~5-15 points instead of 98.  Points come from per-lane
// 16-byte loads of the interleaved copy xy[S][2].  A wave with any lane outside the grid (or non-finite)
// takes the full scan as a whole (uniform branch).  meta = [x0, y0, 1/cell, nx, ny, ...].
constexpr int GRID_CELLS = 65536;   // cells per centerline row, at most
constexpr int GRID_META = 8;        // doubles per row
// K1b on a shared centerline keeps the row's interleaved points in LDS (16 S bytes per wave) only up to this
// many points: 8 KB per wave leaves the kernel its three waves per SIMD; longer tables read the copy in
// global memory (one more trip to L2 per candidate point) -- same index either way
constexpr int GRID_LDS_MAX_S = 512;
struct NearTab {
    const double *boxes;            // [C][NB][4]        block boxes (nearest_index_blocks), or null
    const double *gmeta;            // [C][GRID_META]    grid placement, or null
    const unsigned *gcells;         // [C][GRID_CELLS]   lo | hi << 16
    const double *gxy;              // [C][S][2]         interleaved points
};

// pt(i): point i of the row as a double2 (the interleaved copy in global memory, or a workgroup's copy of it in LDS)
template <class Pt>
MPC_DEV int nearest_index_grid(const DevCfg &c, const double *__restrict__ cl, const double *__restrict__ meta,
                               const unsigned *__restrict__ cells, Pt pt, double px, double py)
{
#pragma clang fp contract(off)
    const double fx = (px - meta[0]) * meta[2], fy = (py - meta[1]) * meta[2];
    const bool in = fx >= 0.0 && fx < meta[3] && fy >= 0.0 && fy < meta[4];
    if (__ballot(!in) != 0ull) return nearest_index(c, cl, px, py);
    const unsigned r = cells[(int)fy * (int)meta[3] + (int)fx];
    const int lo = (int)(r & 0xffffu), hi = (int)(r >> 16);
    const double2 p0 = pt(lo);
    double best = dist2(p0.x, p0.y, px, py);
    int idx = lo;
    // four points per trip (their loads are in flight together); a lane whose range has ended repeats
    // its last point, which the strict "<" ignores
    for (int t = lo + 1; __ballot(t <= hi) != 0ull; t += 4) {
        const int i0 = min(t, hi), i1 = min(t + 1, hi), i2 = min(t + 2, hi), i3 = min(t + 3, hi);
        const double2 q0 = pt(i0), q1 = pt(i1), q2 = pt(i2), q3 = pt(i3);
        const double d0 = dist2(q0.x, q0.y, px, py), d1 = dist2(q1.x, q1.y, px, py);
        const double d2 = dist2(q2.x, q2.y, px, py), d3 = dist2(q3.x, q3.y, px, py);
        const bool l01 = d1 < d0, l23 = d3 < d2;
        const double m01 = l01 ? d1 : d0, m23 = l23 ? d3 : d2;
        const int i01 = l01 ? i1 : i0, i23 = l23 ? i3 : i2;
        const bool lq = m23 < m01;
        const double mq = lq ? m23 : m01;
        const int iq = lq ? i23 : i01;
        const bool lt = mq < best;
        best = lt ? mq : best;
        idx = lt ? iq : idx;
    }
    return idx;
}

// the search the tables of `nt` allow for centerline row `row` (all of them return nearest_index's answer)
MPC_DEV int nearest_lookup(const DevCfg &c, const double *__restrict__ clp, const NearTab &nt, int row,
                           double px, double py)
{
    if (nt.gmeta) {
        const double *meta = nt.gmeta + (size_t)row * GRID_META;
        const unsigned *cells = nt.gcells + (size_t)row * GRID_CELLS;
        const double2 *__restrict__ gp = reinterpret_cast<const double2 *>(nt.gxy + (size_t)row * 2 * (size_t)c.S);
        return nearest_index_grid(c, clp, meta, cells, [=](int i) { return gp[i]; }, px, py);
    }
    if (nt.boxes) {
        const int NB = (c.S - 1 + NEAR_BLK - 1) / NEAR_BLK;
        return nearest_index_blocks(c, clp, nt.boxes + (size_t)row * NB * 4, px, py);
    }
    return nearest_index(c, clp, px, py);
}

struct Geom { double nx_, ny_, px_, py_, qx_, qy_; }; // nearest, previous, next

MPC_DEV void load_geom(const DevCfg &c, const double *__restrict__ cl, int idx, Geom &g)
{
    const int S = c.S;
    const int ip = idx > 0 ? idx - 1 : 0; // car_dynamics.py:183: previous == nearest at index 0
    g.nx_ = cl[idx]; g.ny_ = cl[S + idx];
    g.px_ = cl[ip]; g.py_ = cl[S + ip];
    g.qx_ = cl[idx + 1]; g.qy_ = cl[S + idx + 1];
}

// the same three points out of an interleaved copy of the row
MPC_DEV void load_geom_xy(const double2 *xy, int idx, Geom &g)
{
    const double2 n = xy[idx], p = xy[idx > 0 ? idx - 1 : 0], q = xy[idx + 1];
    g.nx_ = n.x; g.ny_ = n.y; g.px_ = p.x; g.py_ = p.y; g.qx_ = q.x; g.qy_ = q.y;
}

// car_dynamics.py:168-172 with the three possible lowerings of np.mod on an SX
MPC_DEV double wrap_to_pi(const DevCfg &c, double ang)
{
    const double PI = 3.14159265358979323846;
    const double two_pi = 2.0 * PI;
    const double a = ang + PI;
    double m;
    // fmod(a, b) == a exactly for 0 <= a < b: the usual case needs no remainder loop
    const bool plain = a >= 0.0 && a < two_pi;
    if (c.wrap_mode == 1) m = plain ? a : ocml_fmod(a, two_pi);
    else if (c.wrap_mode == 2) m = ocml_remainder(a, two_pi);
    else { m = plain ? a : ocml_fmod(a, two_pi); if (m < 0.0) m += two_pi; }
    return m - PI;
}

// car_dynamics.py:211-228
MPC_DEV void tracking_errors(const DevCfg &c, const Geom &g, double px, double py, double phi,
                             double &cte, double &he, double &pe)
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    cte = (px - g.px_) * (g.ny_ - g.py_) - (py - g.py_) * (g.nx_ - g.px_);
    const double desired = m_atan2(g.qy_ - g.ny_, g.qx_ - g.nx_);
    he = wrap_to_pi(c, desired - phi);
    pe = (px - g.nx_) * (g.qy_ - g.ny_) - (py - g.ny_) * (g.qx_ - g.nx_);
}

// car_dynamics.py:252-257.  GRAD: xb += dL/dx, ub += dL/du.
template <int MODEL, bool GRAD>
MPC_DEV double stage_cost(const DevCfg &c, const Geom &g, const double (&x)[ModelDim<MODEL>::NX],
                          double d, double dl, double (&xb)[ModelDim<MODEL>::NX], double (&ub)[2])
{
#pragma clang fp contract(off)   // fixed roundings: the same bits in every kernel this is inlined into
    double cte, he, pe;
    tracking_errors(c, g, x[0], x[1], x[2], cte, he, pe);
    double sp;
    if (MODEL == PAC) sp = sqrt(x[3] * x[3] + x[4] * x[4]);
    else sp = x[3];
    const double ev = sp - c.v_ref;
    double L = (c.w[0] * ev) * ev;
    L = fma(c.w[1] * cte, cte, L);
    L = fma(c.w[2] * pe, pe, L);
    L = fma(c.w[3] * he, he, L);
    L = fma(c.w[4] * dl, dl, L);
    L = fma(c.w[5] * d, d, L);
    if (GRAD) {
        xb[0] += 2.0 * c.w[1] * cte * (g.ny_ - g.py_) + 2.0 * c.w[2] * pe * (g.qy_ - g.ny_);
        xb[1] += -2.0 * c.w[1] * cte * (g.nx_ - g.px_) - 2.0 * c.w[2] * pe * (g.qx_ - g.nx_);
        xb[2] += -2.0 * c.w[3] * he;
        if (MODEL == PAC) {
            xb[3] += 2.0 * c.w[0] * ev * x[3] / sp;
            xb[4] += 2.0 * c.w[0] * ev * x[4] / sp;
        } else {
            xb[3] += 2.0 * c.w[0] * ev;
        }
        ub[0] += 2.0 * c.w[5] * d;
        ub[1] += 2.0 * c.w[4] * dl;
    }
    return L;
}

// per-stage general constraints (main.py:43-52 or the lane band), value i of the stage
template <int MODEL>
MPC_DEV double stage_constraint(const DevCfg &c, const Geom &g,
                                const double (&x)[ModelDim<MODEL>::NX], int i)
{
#pragma clang fp contract(off)
    if (c.constr_mode == 1) return x[i] * x[i] - c.g_off[i];
    const double wx = g.qx_ - g.nx_, wy = g.qy_ - g.ny_;
    const double pe = (x[0] - g.nx_) * wy - (x[1] - g.ny_) * wx;
    return pe / sqrt(wx * wx + wy * wy); // road.py:77-79
}

template <int MODEL>
MPC_DEV void stage_constraint_adjoint(const DevCfg &c, const Geom &g,
                                      const double (&x)[ModelDim<MODEL>::NX], int i, double yh,
                                      double (&xb)[ModelDim<MODEL>::NX])
{
#pragma clang fp contract(off)
    if (c.constr_mode == 1) { xb[i] += yh * 2.0 * x[i]; return; }
    const double wx = g.qx_ - g.nx_, wy = g.qy_ - g.ny_;
    const double nrm = sqrt(wx * wx + wy * wy);
    xb[0] += yh * wy / nrm;
    xb[1] += -yh * wx / nrm;
}

MPC_DEV void constraint_bounds(const DevCfg &c, int i, double &lb, double &ub)
{
    if (c.constr_mode == 2) { lb = -c.lane_hw; ub = c.lane_hw; }
    else { lb = c.D_lb[i]; ub = c.D_ub[i]; }
}

} // namespace mpc
