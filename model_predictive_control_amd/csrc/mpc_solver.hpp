// mpc_solver.hpp -- K2..K5: the per-agent solver state machine, one WAVEFRONT per agent.
//
// Restates alpaqa's ALMSolver / StructuredPANOCLBFGSSolver as configured at controller.py:27-48
// (algorithm statement and provenance: DESIGN.md).  Every per-agent vector (n = 2N <= 128
// doubles) is a contiguous agent-major row; lane j of the wave owns element j (and j + 64), so a
// row is one coalesced 8n-byte access, vector updates are one instruction for the whole agent and
// inner products are wavefront shuffle reductions.  Control flow is wave-uniform (one agent per
// wave), so nothing diverges.  The L-BFGS history rows of an agent are read ONCE per two-loop
// recursion (K3) -- by LDS-DMA into the wave's LDS slice, or into registers (MC = 20 variant).
#pragma once
#include "mpc_device.hpp"
#include <float.h>

namespace mpc {

// per-agent scalar record: REC doubles, agent-major (integers are stored as exact doubles)
constexpr int REC = 64;
enum {
    R_PSI, R_L, R_GAMMA, R_PHI, R_PSIXH, R_PP, R_GP, R_TAU, R_PSIN, R_LN, R_GAMMAN, R_PSIXHN, R_GPN,
    R_PPN, R_SIGPP, R_EPS, R_HN2, R_HFD, R_GAMMA_TOP, R_DELTA, R_RHO, R_EPS_OLD, R_NE1, R_PS_EPS,
    R_OUT_EPS, R_OUT_DELTA, R_PSI_OUT, R_PSIE,
    R_PHASE, R_K, R_LIDX, R_LFULL, R_NOPROG, R_NJ, R_OUTER, R_FIRST, R_INITRED, R_PENRED,
    R_INNER_TOT, R_INNER_FAIL, R_STATUS, R_NEVALS, R_MAXIT, R_OVERWRITE, R_FALLBACK, R_PS_STATUS,
    R_PS_ITERS, R_OUT_OF_ITER, R_NGRAD, R_LBROWS, R_SPEC, R_SPEC_GAMMA, R_NSPEC, R_NSPEC_USED, R_NCOST,
    // the inner solve in progress (smallest stop measure seen, evaluation count at its start) and the memo of
    // the last one that failed and was backtracked over without constraints (see PH_OUTER_BEGIN)
    R_RUN_MINEPS, R_RUN_EV0, R_MEMO_STATUS, R_MEMO_ITERS, R_MEMO_EVALS, R_MEMO_MINEPS, R_MEMO_EPS,
    // persistent kernel's lookahead (mpc_solo.hpp): candidate evaluations executed beside requested ones, requests served
    // from them without a trip (plain double counters)
    R_LA_EVALS, R_LA_HITS, R_USED
};
static_assert(R_USED <= REC, "record too small");
// The integers of the record (phase, counters, flags: all >= 0) are stored as the double 2^52 + k, whose low dword IS
// k: a read is one v_readlane, a write one v_cndmask of the low dword -- as (double)k they cost two v_readlane, a
// v_cvt_i32_f64 and a v_readfirstlane per read, and the state machine reads ~30 of them per step (round 3: the step
// kernel is issue-bound at four waves per SIMD).  R_NGRAD .. R_NCOST are plain double counters (they are only added to).
__host__ __device__ constexpr bool rec_is_int(int s)
{
    return (s >= R_PHASE && s <= R_OUT_OF_ITER) || s == R_SPEC || (s >= R_RUN_EV0 && s <= R_MEMO_EVALS);
}
__device__ __forceinline__ double rec_int(int k) { return __hiloint2double(0x43300000, k); }
__device__ __forceinline__ int rec_int_of(double v) { return __double2loint(v); }

enum Phase {
    PH_DONE = 0,
    PH_W_INIT_H = 1, PH_W_INIT_X, PH_W_DL, PH_W_HEUR, PH_W_HESS, PH_W_LS_G, PH_W_LS_C, // wait for K1
    PH_OUTER_BEGIN = 16, PH_TOP, PH_AFTER_DL, PH_LS_INIT, PH_LS_TRIAL, PH_INNER_EXIT   // internal
};
enum { ST_UNKNOWN = 0, ST_CONVERGED = 1, ST_MAXTIME = 2, ST_MAXITER = 3, ST_NOTFINITE = 4,
       ST_NOPROGRESS = 5 };
enum { REQ_NONE = 0, REQ_GRAD = 1, REQ_COST = 2, REQ_SPEC = 4, // bit set; SPEC = gradient on channel 2
       REQ_CHAIN = 8 };  // with REQ_GRAD: the gradient at a line-search trial point, whose follow-up K1c may run itself
// a work-list entry is the agent id, plus CH2_BIT when the evaluation runs on the speculative
// channel (input row xe2, gradient row ge2, nothing else written), plus CHAIN_BIT on the gradient list when
// it is the gradient at a line-search trial point (PH_LS_TRIAL -> PH_W_LS_G).  What the agent does with that
// gradient is fixed and elementwise -- prox step at the trial point, request the cost there, speculate
// (PH_W_LS_G) -- and is done for it by one THREAD instead of one wavefront: the next step-kernel launch carries
// extra workgroups (chain_block) that walk the gradient slots of the finished round, 64 per workgroup, and do
// PH_W_LS_G for the entries with this bit -- a third of all agent-steps at a fifth of the instructions.
constexpr int CH2_BIT = 1 << 30;
constexpr int CHAIN_BIT = 1 << 29;
constexpr int AGENT_MASK = CHAIN_BIT - 1;
// A chain_block runs beside the wave-per-agent blocks of the same launch, which must not pick up the agent it has
// just moved to PH_W_LS_C (its cost evaluation has not run yet): it writes the phase with a tag of the launch's
// parity, PH_W_LS_C + 64 (1 + round % 2); the blocks of a launch leave agents with their own launch's tag alone
// and take those with the other one -- written a round ago, the evaluation done -- and the tag is dropped.
constexpr int PH_MASK = 63;
__device__ __forceinline__ int chain_tag(int par) { return 64 * (1 + par); }

struct Workspace {
    const double *x0;                              // [B][nx]   caller's buffer
    double *xo;                                    // [B][n]    caller's U (warm start in, solution out)
    double *xk, *gk, *q, *xn, *xe, *ge;            // [B][n]
    double *xe2, *ge2;                             // [B][n]    speculative channel (point, gradient)
    double *S, *Y;                                 // [B][M][n] L-BFGS history
    double *y;                                     // [B][m]    caller's lambda (in/out)
    double *Sig, *Sig_old, *e1, *e2, *yhx, *yhxn, *yhe; // [B][m]
    double *rec;                                   // [B][REC]
    int *lists;                                    // [2 buffers][2 kinds][Bp] agent ids
    int *counts;                                   // [2 buffers][4]
    unsigned long long *totals;                    // [16] gradient evals, cost evals, history pairs read, harness, spec issued/used,
                                                   //      unfinished agents, lookahead evaluations / hits
    int *solo_ctr;                                 // [groups][2] persistent kernel: claim counter, list length
    // K1 scratch, slot-indexed SoA with stride Bp + 64 (see mpc_eval.hpp)
    double *trajx;                                 // [(N+1)*nx][St] x_0 .. x_N
    double *useq;                                  // [2N][St]       control sequence
    double *stage_L;                               // [N][St]        stage costs (+ ALM terms)
    double *jac;                                   // [N*JS][St]     dL/dx, dL/du, stage sensitivities
    int *agent_of;                                 // [St]           agent of a slot (-1: none)
    int *arrive;                                   // [St / 64]      stage blocks done per slot block (K1c inside K1b), or null
    const double *cl;                              // [C][2S]
    const int *cl_index;                           // [B] or null
    NearTab near;                                  // tables of the pruned nearest-point searches (all null: full scan)
    double *psi_direct;                            // direct-mode K1 output (standalone evaluation)
    double *ws_xe, *ws_ge, *ws_yhe, *ws_Sig;       // the workspace's own rows (xe/ge/yhe/Sig may alias caller buffers)
    int B, Bp;                                     // agents of this view, rounded up to 64
    int St;                                        // stride of the slot-indexed K1 scratch
    int Ls;                                        // stride between the two work lists
};

#ifdef MPC_DEV_STAMP
// (timing experiments, never in the product build: -DMPC_DEV_STAMP=1 K1a (two lanes per request), 2 the fused K1b+K1c kernel,
// 3 the step kernel, 4 K1a of the Pacejka model (four lanes per request), 5 the persistent kernel (per agent), 6 K1b (stage_kernel))
// per wave of the last launch: start, end (100 MHz clock), HW_ID, XCC_ID | counters
constexpr int DEV_STAMPS = 65536;
__device__ long long g_dev_stamps[4 * DEV_STAMPS];
struct DevStamp {
    long long t0; int idx; int nfall = 0, nmid = 0, nslow = 0;
    __device__ DevStamp(int i) : t0(__builtin_amdgcn_s_memrealtime()), idx(i) {}
    __device__ ~DevStamp()
    {
        if ((threadIdx.x & 63) == 0 && idx < DEV_STAMPS) {
            g_dev_stamps[4 * idx] = t0; g_dev_stamps[4 * idx + 1] = __builtin_amdgcn_s_memrealtime();
            g_dev_stamps[4 * idx + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
            g_dev_stamps[4 * idx + 3] = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15)  // XCC_ID
                                        | ((long long)nfall << 8) | ((long long)nmid << 16) | ((long long)nslow << 24);
        }
    }
};
#endif
// ------------------------------------------------------------------ wavefront helpers
__device__ __forceinline__ double rdlane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// Wavefront reductions on DPP row operations (no LDS round trip): four exchange steps give every
// lane its 16-lane row total, the four row totals are read through SGPRs and added in a fixed
// order, so all lanes hold the same bits.  Must be called with all 64 lanes active.
template <int CTRL>
__device__ __forceinline__ double dpp_xchg(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#ifdef MPC_SCAN_SUM
// Wavefront sum as a DPP scan: four row_shr steps leave each 16-lane row's total in its last lane,
// row_bcast:15 / row_bcast:31 (GFX9 cross-row DPP) carry the totals upwards, lane 63 ends up with
// the sum of all 64 lanes and is read through SGPRs -- 6 x (2 v_mov_dpp + 1 v_add_f64) + 2
// v_readlane, no LDS, no VGPR<->SGPR juggling for the cross-row part.  All lanes active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double scan_sum(double v) // lane 63 holds the wave total afterwards
{
    v += dpp_take<0x111, 0xf>(v); // row_shr:1 (lanes shifted in from outside the row read 0)
    v += dpp_take<0x112, 0xf>(v); // row_shr:2
    v += dpp_take<0x114, 0xf>(v); // row_shr:4
    v += dpp_take<0x118, 0xf>(v); // row_shr:8
    v += dpp_take<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return rdlane(scan_sum(v), 63); }
#else
__device__ __forceinline__ double row_sum16(double v)
{
    v += dpp_xchg<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_xchg<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_xchg<0x141>(v); // row_half_mirror
    v += dpp_xchg<0x140>(v); // row_mirror
    return v;
}
__device__ __forceinline__ double scan_sum(double v) { return row_sum16(v); }
// The four row totals are added as (r0 + r1) + (r2 + r3).  A vector of n <= 48 (<= 32) elements leaves
// row 3 (rows 2 and 3) all zero: their totals are skipped -- adding an exact zero changes no bit -- and
// with them two (four) SGPR reads and an addition per sum (NROWS is what the caller knows about n).
template <int NROWS = 4>
__device__ __forceinline__ double cross_rows(double v)
{
    if (NROWS <= 2) return rdlane(v, 0) + rdlane(v, 16);
    if (NROWS == 3) return (rdlane(v, 0) + rdlane(v, 16)) + rdlane(v, 32);
    return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
}
__device__ __forceinline__ double wave_sum(double v)
{
    v = row_sum16(v);
    return cross_rows<4>(v);
}
#endif
// reciprocal by v_rcp_f64 + two Newton steps (<= 1 ulp): the IEEE division sequence costs ~6x more
// issue slots, and the two-loop does one per history pair
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
// finite-difference step of the Hessian-vector products: cbrt(eps) (1 + ||x||); one function for every place
// that forms it (the state machine, the speculation, K1c's chained step) so that they agree bit for bit
__device__ __forceinline__ double fd_step(double xx)
{
#pragma clang fp contract(off)
    return cbrt(DBL_EPSILON) * (1.0 + sqrt(xx));
}
#ifdef MPC_SCAN_SUM
__device__ __forceinline__ void wave_sum2(double &a, double &b)
{
    a = scan_sum(a); b = scan_sum(b);
    a = rdlane(a, 63); b = rdlane(b, 63);
}
__device__ __forceinline__ void wave_sum3(double &a, double &b, double &c)
{
    a = scan_sum(a); b = scan_sum(b); c = scan_sum(c);
    a = rdlane(a, 63); b = rdlane(b, 63); c = rdlane(c, 63);
}
__device__ __forceinline__ double wave_sum_n(double v, int) { return wave_sum(v); }
__device__ __forceinline__ void wave_sum2_n(double &a, double &b, int) { wave_sum2(a, b); }
#else
__device__ __forceinline__ void wave_sum2(double &a, double &b)
{
    a = row_sum16(a); b = row_sum16(b);
    a = cross_rows<4>(a); b = cross_rows<4>(b);
}
// sums of vectors whose elements beyond n are exact zeros (one element per lane: NE = 1), n wave-uniform
__device__ __forceinline__ double wave_sum_n(double v, int n)
{
    v = row_sum16(v);
    return n <= 32 ? cross_rows<2>(v) : n <= 48 ? cross_rows<3>(v) : cross_rows<4>(v);
}
__device__ __forceinline__ void wave_sum2_n(double &a, double &b, int n)
{
    a = row_sum16(a); b = row_sum16(b);
    if (n <= 32) { a = cross_rows<2>(a); b = cross_rows<2>(b); }
    else if (n <= 48) { a = cross_rows<3>(a); b = cross_rows<3>(b); }
    else { a = cross_rows<4>(a); b = cross_rows<4>(b); }
}
__device__ __forceinline__ void wave_sum3(double &a, double &b, double &c)
{
    a = row_sum16(a); b = row_sum16(b); c = row_sum16(c);
    a = (rdlane(a, 0) + rdlane(a, 16)) + (rdlane(a, 32) + rdlane(a, 48));
    b = (rdlane(b, 0) + rdlane(b, 16)) + (rdlane(b, 32) + rdlane(b, 48));
    c = (rdlane(c, 0) + rdlane(c, 16)) + (rdlane(c, 32) + rdlane(c, 48));
}
#endif
__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_xchg<0xB1>(v)); v = fmax(v, dpp_xchg<0x4E>(v));
    v = fmax(v, dpp_xchg<0x141>(v)); v = fmax(v, dpp_xchg<0x140>(v));
    return fmax(fmax(rdlane(v, 0), rdlane(v, 16)), fmax(rdlane(v, 32), rdlane(v, 48)));
}

// Wave-uniform scalars.  Every assignment goes through v_readfirstlane, so the value lives in
// SGPRs (VALU results would otherwise pin a VGPR pair per scalar next to the cached history rows).
__device__ __forceinline__ double rfl(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
struct RecD { // a double scalar of the agent record, held in lane `slot` of the record register
    double &rv; const int lane, slot;
    __device__ __forceinline__ operator double() const { return rdlane(rv, slot); }
    __device__ __forceinline__ RecD &operator=(double x) { rv = lane == slot ? x : rv; return *this; }
    __device__ __forceinline__ RecD &operator=(const RecD &o) { return *this = (double)o; }
    __device__ __forceinline__ RecD &operator*=(double x) { return *this = (double)*this * x; }
    __device__ __forceinline__ RecD &operator/=(double x) { return *this = (double)*this / x; }
};
struct RecI { // an integer scalar of the record (stored as 2^52 + k: rec_int)
    double &rv; const int lane, slot;
    __device__ __forceinline__ operator int() const { return __builtin_amdgcn_readlane(__double2loint(rv), slot); }
    __device__ __forceinline__ RecI &operator=(int x)
    {
        const int lo = __double2loint(rv);
        rv = __hiloint2double(__double2hiint(rv), lane == slot ? x : lo);
        return *this;
    }
    __device__ __forceinline__ RecI &operator=(const RecI &o) { return *this = (int)o; }
    __device__ __forceinline__ RecI &operator+=(int x) { return *this = (int)*this + x; }
    __device__ __forceinline__ RecI &operator|=(int x) { return *this = (int)*this | x; }
    __device__ __forceinline__ RecI &operator++(int) { return *this = (int)*this + 1; }
};

// The phase word and the iteration counter are read at every turn of the state machine's loop (two v_readlane, a
// conversion and a v_readfirstlane each time): they are taken out of the record once per step and put back at its end
// (round 3, with the kernel issue-bound at four waves per SIMD: step kernel 65.2 -> 62.9 ms per solve, 428.4 k -> 436.2 k
// solves/s; the same with a dozen more scalars costs registers the kernel does not have: profiles/r03_experiments.txt).
struct LocI {
    int v; const int slot;
    __device__ __forceinline__ LocI(double rv, int s) : v(__builtin_amdgcn_readlane(__double2loint(rv), s)), slot(s) {}
    __device__ __forceinline__ operator int() const { return v; }
    __device__ __forceinline__ LocI &operator=(int x) { v = x; return *this; }
    __device__ __forceinline__ LocI &operator++(int) { v = v + 1; return *this; }
    __device__ __forceinline__ void put(double &rv, int lane) const
    { rv = __hiloint2double(__double2hiint(rv), lane == slot ? v : __double2loint(rv)); }
};

// a row of n <= 64*NE doubles spread over the wave: element e of lane l is index l + 64 e
template <int NE> struct Row { double v[NE]; };

template <int NE>
__device__ __forceinline__ Row<NE> ldrow(const double *__restrict__ p, int n, int lane)
{
    Row<NE> r;
#pragma unroll
    for (int e = 0; e < NE; e++) { const int j = lane + 64 * e; r.v[e] = j < n ? p[j] : 0.0; }
    return r;
}
template <int NE>
__device__ __forceinline__ void strow(double *__restrict__ p, int n, int lane, const Row<NE> &r)
{
#pragma unroll
    for (int e = 0; e < NE; e++) { const int j = lane + 64 * e; if (j < n) p[j] = r.v[e]; }
}

__device__ __forceinline__ double prox_p(const DevCfg &c, int par, double x, double g, double gamma)
{
#pragma clang fp contract(off)   // fixed roundings: the step kernel and the persistent kernel must agree bit for bit
    // comparison-selects, not fmin/fmax: a NaN gradient must stay a NaN step (Eigen's cwiseMax/cwiseMin
    // in alpaqa's projection keep it too), so that ||p||/gamma is NaN and the stop test says NotFinite
    const double lo = c.u_lb[par] - x, hi = c.u_ub[par] - x;
    double p = -gamma * g;
    p = p < lo ? lo : p;
    return hi < p ? hi : p;
}
__device__ __forceinline__ bool in_J(const DevCfg &c, int par, double x, double g, double gamma)
{
#pragma clang fp contract(off)   // fixed roundings: the step kernel and the persistent kernel must agree bit for bit
    const double gd = x - gamma * g;
    return !(gd < c.u_lb[par] || c.u_ub[par] < gd);
}

// K2: forward-backward step of (xb, gb) with step gamma: xhat -> xe row, returns ||p||^2, g'p
template <int NE>
__device__ __forceinline__ void prox_to_xe(const DevCfg &c, double *__restrict__ xe, int n, int lane,
                                           const Row<NE> &xb, const Row<NE> &gb, double gamma,
                                           double &pp, double &gp)
{
#pragma clang fp contract(off)   // fixed roundings: the step kernel and the persistent kernel must agree bit for bit
    Row<NE> xh;
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int e = 0; e < NE; e++) {
        const double p = prox_p(c, lane & 1, xb.v[e], gb.v[e], gamma);
        xh.v[e] = xb.v[e] + p;
        if (lane + 64 * e < n) { a = fma(p, p, a); b = fma(gb.v[e], p, b); }
    }
    strow<NE>(xe, n, lane, xh);
    wave_sum2_n(a, b, n);
    pp = a; gp = b;
}

// The S and Y history blocks of one agent (M n doubles each, contiguous) copied to LDS by the
// LDS-DMA path: 16 B per lane and instruction, no VGPRs, completion tracked by vmcnt.
// (Mn = P n: the LDS copy holds the ring slots 0 .. P - 1 of each block, P <= M -- the rest of a long history
// is read from global memory by the two-loop)
__device__ __forceinline__ void hist_dma(const double *__restrict__ gS, const double *__restrict__ gY,
                                         double *hist, int Mn, int rows_n, int lane)
{
    rows_n = rows_n < Mn ? rows_n : Mn;
    // only the ring slots in use are fetched: the first `rows_n` doubles of each block (a ring that
    // has not wrapped yet holds its pairs in slots 0 .. lidx-1; the history is flushed whenever the
    // step size changes, so on average it is far from full)
    const int bytes = rows_n * 8;
    for (int off = 0; off < bytes; off += 1024) {
        const int my = off + lane * 16;
        if (my < bytes) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)((const char *)gS + my),
                (__attribute__((address_space(3))) void *)((char *)hist + off), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)((const char *)gY + my),
                (__attribute__((address_space(3))) void *)((char *)(hist + Mn) + off), 16, 0, 0);
        }
    }
}
__device__ __forceinline__ void hist_wait()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// K3: masked L-BFGS two-loop (alpaqa LBFGS::apply(q, -1, J)) for one agent held by one wave.
// rho is recomputed on J, pairs with rho <= 0 are skipped, H0 = s'y / y'y of the newest valid
// pair.  MC > 0: the cnt <= MC history rows are loaded once into registers and serve both loops.
// MC < 0: Sa / Ya point at the wave's LDS copy of the agent's whole history (hist_dma), rows are
// read from there with the next pair requested while the current one is being used.
template <int NE, int MC>
__device__ __forceinline__ bool lbfgs_two_loop(const DevCfg &c, const double *__restrict__ Sa,
                                               const double *__restrict__ Ya, int n, int lane,
                                               const bool (&inj)[NE], int lidx, int lfull,
                                               Row<NE> &q, int &rows_read, int P = 1 << 30,
                                               const double *__restrict__ Sg = nullptr,
                                               const double *__restrict__ Yg = nullptr)
{
#pragma clang fp contract(off)   // fixed roundings: the step kernel and the persistent kernel must agree bit for bit
    const int M = c.M;
    const int cnt = lfull ? M : lidx;
    if (cnt == 0) return false;
    rows_read += cnt;
    double alpha_v = 0.0, rho_v = -1.0; // lane t keeps alpha_t / rho_t
    double h0 = -1.0;
    constexpr int MCC = MC > 0 ? MC : 1;
    Row<NE> sc[MCC], yc[MCC];
    if (MC > 0) {
#pragma unroll
        for (int t = 0; t < MCC; t++) {
            if (t < cnt) {
                int i = lidx - 1 - t; if (i < 0) i += M;
                sc[t] = ldrow<NE>(Sa + (size_t)i * n, n, lane);
                yc[t] = ldrow<NE>(Ya + (size_t)i * n, n, lane);
#pragma unroll
                for (int e = 0; e < NE; e++) if (!inj[e]) { sc[t].v[e] = 0.0; yc[t].v[e] = 0.0; }
            }
        }
    }
    auto first_loop = [&](int t, const Row<NE> &s, const Row<NE> &y) {
        double sy = 0.0, sq = 0.0;
#pragma unroll
        for (int e = 0; e < NE; e++) { sy = fma(s.v[e], y.v[e], sy); sq = fma(s.v[e], q.v[e], sq); }
        wave_sum2_n(sy, sq, n);
        const double rho = fast_rcp(sy);
        if (!(rho > 0.0)) return;          // lane t keeps rho_t = -1
        const double al = rho * sq;
        if (lane == t) { rho_v = rho; alpha_v = al; }
#pragma unroll
        for (int e = 0; e < NE; e++) q.v[e] = fma(-al, y.v[e], q.v[e]);
        if (h0 < 0.0) { // y'y is only needed once, for H0 of the newest valid pair (wave-uniform branch)
            double yy = 0.0;
#pragma unroll
            for (int e = 0; e < NE; e++) yy = fma(y.v[e], y.v[e], yy);
            h0 = 1.0 / (rho * wave_sum_n(yy, n));
        }
    };
    auto second_loop = [&](int t, const Row<NE> &s, const Row<NE> &y) {
        const double rho = rdlane(rho_v, t);
        if (!(rho > 0.0)) return;
        double yq = 0.0;
#pragma unroll
        for (int e = 0; e < NE; e++) yq = fma(y.v[e], q.v[e], yq);
        yq = wave_sum_n(yq, n);
        const double ab = rdlane(alpha_v, t) - rho * yq;
#pragma unroll
        for (int e = 0; e < NE; e++) q.v[e] = fma(ab, s.v[e], q.v[e]);
    };
    auto load_masked = [&](int t, Row<NE> &s, Row<NE> &y) {
        int i = lidx - 1 - t; if (i < 0) i += M;
        if (MC < 0 && i < P) {                           // (uniform) the wave's LDS copy holds ring slots 0 .. P - 1
            typedef const __attribute__((address_space(3))) double lds_cd;
            lds_cd *ls = (lds_cd *)Sa + i * n, *ly = (lds_cd *)Ya + i * n;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const int j = lane + 64 * e < n ? lane + 64 * e : 0;
                s.v[e] = ls[j]; y.v[e] = ly[j];
            }
        } else if (MC < 0) {                             // a slot beyond the copy: from the history in global memory
            s = ldrow<NE>(Sg + (size_t)i * n, n, lane);
            y = ldrow<NE>(Yg + (size_t)i * n, n, lane);
        } else {
            s = ldrow<NE>(Sa + (size_t)i * n, n, lane);
            y = ldrow<NE>(Ya + (size_t)i * n, n, lane);
        }
#pragma unroll
        for (int e = 0; e < NE; e++) if (!inj[e]) { s.v[e] = 0.0; y.v[e] = 0.0; }
    };
    if (MC > 0) {
#pragma unroll
        for (int t = 0; t < MCC; t++) if (t < cnt) first_loop(t, sc[t], yc[t]);
    } else if (MC < 0) {
        Row<NE> s, y;
        load_masked(0, s, y);
        for (int t = 0; t < cnt; t++) {
            Row<NE> s2 = s, y2 = y;
            if (t + 1 < cnt) load_masked(t + 1, s2, y2);
            first_loop(t, s, y);
            s = s2; y = y2;
        }
    } else {
        for (int t = 0; t < cnt; t++) { Row<NE> s, y; load_masked(t, s, y); first_loop(t, s, y); }
    }
    if (h0 < 0.0) return false;
#pragma unroll
    for (int e = 0; e < NE; e++) if (inj[e]) q.v[e] *= h0;
    if (MC > 0) {
#pragma unroll
        for (int t = MCC - 1; t >= 0; t--) if (t < cnt) second_loop(t, sc[t], yc[t]);
    } else if (MC < 0) {
        Row<NE> s, y;
        load_masked(cnt - 1, s, y);
        for (int t = cnt - 1; t >= 0; t--) {
            Row<NE> s2 = s, y2 = y;
            if (t > 0) load_masked(t - 1, s2, y2);
            second_loop(t, s, y);
            s = s2; y = y2;
        }
    } else {
        rows_read += cnt; // rows are read a second time
        for (int t = cnt - 1; t >= 0; t--) { Row<NE> s, y; load_masked(t, s, y); second_loop(t, s, y); }
    }
    return true;
}

// The line-search trial point for step tau (PH_LS_TRIAL) and the point of the speculative Hessian-vector gradient
// (speculate / PH_AFTER_DL): one function each for the state machine and for the persistent kernel's lookahead
// (mpc_solo.hpp), which forms the SAME points ahead of the state machine and must get the same bits.
template <int NE>
__device__ __forceinline__ Row<NE> trial_point(const DevCfg &c, int par, const Row<NE> &x, const Row<NE> &g,
                                               const Row<NE> &qv, double gamma, double tau, bool fallback)
{
#pragma clang fp contract(off)
    Row<NE> r;
#pragma unroll
    for (int e = 0; e < NE; e++) {
        const double p = prox_p(c, par, x.v[e], g.v[e], gamma);
        if (fallback) r.v[e] = x.v[e] + p;
        else if (tau == 1.0) r.v[e] = x.v[e] + qv.v[e];
        else r.v[e] = x.v[e] + (1.0 - tau) * p + tau * qv.v[e];
    }
    return r;
}
// returns |J| (the point exists when 0 < |J| < n); all 64 lanes must call it
template <int NE>
__device__ __forceinline__ int spec_point(const DevCfg &c, int par, int n, int lane, const Row<NE> &xn, const Row<NE> &ge,
                                          double gm, Row<NE> &out)
{
#pragma clang fp contract(off)
    Row<NE> qv;
    double cntJ = 0.0, xx = 0.0;
#pragma unroll
    for (int e = 0; e < NE; e++) {
        const bool valid = lane + 64 * e < n;
        const bool in = in_J(c, par, xn.v[e], ge.v[e], gm);
        qv.v[e] = in ? 0.0 : prox_p(c, par, xn.v[e], ge.v[e], gm);
        if (valid) { cntJ += in ? 1.0 : 0.0; xx += xn.v[e] * xn.v[e]; }
    }
    wave_sum2_n(cntJ, xx, n);
    const int nj = (int)cntJ;
    if (nj > 0 && nj < n) {
        const double h = fd_step(xx);
#pragma unroll
        for (int e = 0; e < NE; e++) out.v[e] = xn.v[e] + h * qv.v[e];
    }
    return nj;
}

// K5 helper: alpaqa detail::update_penalty_weights (per-constraint factors), lanes stride over m
__device__ __forceinline__ void update_penalty(const DevCfg &c, const Workspace &w, size_t am, int lane,
                                               double Delta, int first, double ne1)
{
#pragma clang fp contract(off)   // fixed roundings: the step kernel and the persistent kernel must agree bit for bit
    for (int k = lane; k < c.m; k += 64) {
        const double so = w.Sig_old[am + k];
        double sv = so;
        if (!(ne1 <= c.alm_delta)) {
            const double e = fabs(w.e1[am + k]), eo = fabs(w.e2[am + k]);
            if (first || e > c.theta * eo) sv = fmin(c.Sigma_max, fmax(Delta * e / ne1, 1.0) * so);
        }
        w.Sig[am + k] = sv;
    }
}

// The solver state machine of agent `a`, executed by one wave.  Returns the evaluation the agent
// now waits for (REQ_GRAD / REQ_COST) or REQ_NONE when it is finished.
// what a step needs from memory before it can decide anything: the scalar record and the five rows
// nearly every phase touches.  Loaded one agent ahead (software pipelining across the agents a wave
// walks), so the memory round trip of agent i+1 overlaps the work of agent i.
template <int NE> struct AgentIn { double rv; Row<NE> X, G, GE, Q, XN, GE2; bool has_q; };

// `ph` = the phase the agent waits in (wave-uniform; the step kernel has it from its one look at the
// workgroup's phase words) selects the rows that phase -- and the chain of internal phases behind it -- reads:
//   W_LS_G  xn, ge                      (prox step at the trial point, speculation)
//   W_LS_C  xk, gk, xn, ge, ge2         (accept: L-BFGS pair, next direction; q only if the trial FAILS: re-read there)
//   W_HESS  xk, gk, ge, q               W_INIT_H  xk, ge          W_INIT_X  xk, ge, q
//   W_DL, W_HEUR  xk, gk, ge, ge2       OUTER_BEGIN  nothing (it reads the caller's U itself)
// Rows not fetched are zeros.  ph < 0 (the persistent kernel, which learns the phase from the record it
// loads here): all of them.  Fetching six rows whatever the phase was 2.4 KB per agent-step; 1.6 KB on average now.
template <int NE>
__device__ __forceinline__ AgentIn<NE> load_agent(const DevCfg &c, const Workspace &w, int a, int lane, int ph = -1)
{
    AgentIn<NE> in;
    const int n = c.n;
    const size_t an = (size_t)a * n;
    in.rv = w.rec[(size_t)a * REC + lane];
    const bool all = ph < 0;
    const bool nX = all || (ph != PH_W_LS_G && ph != PH_OUTER_BEGIN);
    const bool nG = all || ph == PH_W_LS_C || ph == PH_W_HESS || ph == PH_W_DL || ph == PH_W_HEUR;
    const bool nGE = all || ph != PH_OUTER_BEGIN;
    const bool nQ = all || ph == PH_W_HESS || ph == PH_W_INIT_X;
    const bool nXN = all || ph == PH_W_LS_G || ph == PH_W_LS_C;
    const bool nGE2 = all || ph == PH_W_LS_C || ph == PH_W_DL || ph == PH_W_HEUR;
    Row<NE> z;
#pragma unroll
    for (int e = 0; e < NE; e++) z.v[e] = 0.0;
    in.X = nX ? ldrow<NE>(w.xk + an, n, lane) : z; in.G = nG ? ldrow<NE>(w.gk + an, n, lane) : z;
    in.GE = nGE ? ldrow<NE>(w.ge + an, n, lane) : z; in.Q = nQ ? ldrow<NE>(w.q + an, n, lane) : z;
    in.XN = nXN ? ldrow<NE>(w.xn + an, n, lane) : z;
    in.GE2 = nGE2 ? ldrow<NE>(w.ge2 + an, n, lane) : z;
    in.has_q = nQ;
    return in;
}

// HASM = false: the caller knows that the problem has no constraints (m == 0) -- every multiplier / penalty loop
// and the eight pointers behind them drop out of the code (fewer live scalar registers in the round path's kernel).
template <int NE, int MC, bool HASM = true>
__device__ int advance_agent(const DevCfg &c, const Workspace &w, int a, int lane, const AgentIn<NE> &in,
                             double *hist, bool hist_ready, bool allow_spec = true, bool allow_chain = false,
                             int P = 1 << 30)
{
#pragma clang fp contract(off)   // fixed roundings: the step kernel and the persistent kernel must agree bit for bit
    const int n = c.n, m = HASM ? c.m : 0;
    if (P > c.M) P = c.M;                                // ring slots of the history that the LDS copy holds (MC < 0)
    const size_t an = (size_t)a * n, am = (size_t)a * m;
    double *recp = w.rec + (size_t)a * REC;
    // The ~50 per-agent scalars live in LDS for the duration of the step (wave-uniform values would
    // otherwise each occupy a VGPR pair next to the cached history rows).
    // The record and the four rows nearly every phase needs are requested together (one memory
    // round trip); from here on X, G, GE, Q are the register copies of xk, gk, ge, q and are kept
    // coherent with memory, so a chain of phases never re-reads a row it has just written.
    double rv = in.rv;
    Row<NE> X = in.X, G = in.G, GE = in.GE, Q = in.Q, XN = in.XN;
    if (__builtin_amdgcn_readlane(__double2loint(rv), R_PHASE) == PH_DONE) return REQ_NONE;
    // The ~50 per-agent scalars stay where they arrive: slot s of the record in lane s of `rv`.
    // A scalar is read with v_readlane when a phase needs it and written back into its lane;
    // a phase touches a handful of them, so nothing is unpacked or repacked wholesale.
    RecD psie{rv, lane, R_PSIE};
    RecD psik{rv, lane, R_PSI};
    RecD Lk{rv, lane, R_L};
    RecD gamma{rv, lane, R_GAMMA};
    RecD phik{rv, lane, R_PHI};
    RecD psixh{rv, lane, R_PSIXH};
    RecD pp{rv, lane, R_PP};
    RecD gp{rv, lane, R_GP};
    RecD tau{rv, lane, R_TAU};
    RecD psin{rv, lane, R_PSIN};
    RecD Ln{rv, lane, R_LN};
    RecD gamman{rv, lane, R_GAMMAN};
    RecD psixhn{rv, lane, R_PSIXHN};
    RecD gpn{rv, lane, R_GPN};
    RecD ppn{rv, lane, R_PPN};
    RecD sigpp{rv, lane, R_SIGPP};
    RecD eps{rv, lane, R_EPS};
    RecD hn2{rv, lane, R_HN2};
    RecD hfd{rv, lane, R_HFD};
    RecD gamma_top{rv, lane, R_GAMMA_TOP};
    RecD Delta{rv, lane, R_DELTA};
    RecD rho_alm{rv, lane, R_RHO};
    RecD eps_old{rv, lane, R_EPS_OLD};
    RecD ne1{rv, lane, R_NE1};
    RecD ps_eps{rv, lane, R_PS_EPS};
    RecD out_eps{rv, lane, R_OUT_EPS};
    RecD out_delta{rv, lane, R_OUT_DELTA};
    RecD psi_out{rv, lane, R_PSI_OUT};
    LocI phase(rv, R_PHASE);
    LocI k(rv, R_K);
    RecI lidx{rv, lane, R_LIDX};
    RecI lfull{rv, lane, R_LFULL};
    RecI noprog{rv, lane, R_NOPROG};
    RecI nJ{rv, lane, R_NJ};
    RecI outer{rv, lane, R_OUTER};
    RecI first{rv, lane, R_FIRST};
    RecI init_red{rv, lane, R_INITRED};
    RecI pen_red{rv, lane, R_PENRED};
    RecI inner_tot{rv, lane, R_INNER_TOT};
    RecI inner_fail{rv, lane, R_INNER_FAIL};
    RecI status{rv, lane, R_STATUS};
    RecI nevals{rv, lane, R_NEVALS};
    RecI max_it{rv, lane, R_MAXIT};
    RecI overwrite{rv, lane, R_OVERWRITE};
    RecI fallback{rv, lane, R_FALLBACK};
    RecI ps_status{rv, lane, R_PS_STATUS};
    RecI ps_iters{rv, lane, R_PS_ITERS};
    RecI out_of_iter{rv, lane, R_OUT_OF_ITER};
    RecI spec{rv, lane, R_SPEC};
    RecD spec_gamma{rv, lane, R_SPEC_GAMMA};
    RecD run_mineps{rv, lane, R_RUN_MINEPS};
    RecI run_ev0{rv, lane, R_RUN_EV0};
    RecI memo_status{rv, lane, R_MEMO_STATUS};
    RecI memo_iters{rv, lane, R_MEMO_ITERS};
    RecI memo_evals{rv, lane, R_MEMO_EVALS};
    RecD memo_mineps{rv, lane, R_MEMO_MINEPS};
    RecD memo_eps{rv, lane, R_MEMO_EPS};
    if (phase > PH_MASK) phase = (int)phase & PH_MASK;   // (a chain_block's launch tag: see chain_tag)
    double t_pp, t_gp;
    int lb_rows = 0, n_grad = 0;
    int req = REQ_NONE;
    const int par = lane & 1;

    int n_spec = 0, n_used = 0;
    bool hist_landed = false; // the LDS copy of the history has been waited for in this step
    // Speculation: while the cost at xhat(x+) is being evaluated, the gradient the NEXT iteration
    // needs for its Hessian-vector product (at x+ + h q_J, PH_AFTER_DL) is evaluated as well, on the
    // second channel, assuming x+ is accepted with step gm.  Same formulas as PH_AFTER_DL on the same
    // inputs, so the point -- and the gradient -- are bit-identical when the assumption holds.
    auto speculate = [&](double gm) {
        if (c.no_spec || !allow_spec) { spec = 0; return; }
        Row<NE> xh;
        const int nj = spec_point<NE>(c, par, n, lane, XN, GE, gm, xh);
        spec = 0;
        if (nj > 0 && nj < n) {
            strow<NE>(w.xe2 + an, n, lane, xh);
            spec = 1; spec_gamma = gm;
            req |= REQ_SPEC; n_spec = 1;
        }
    };

    while (req == REQ_NONE && phase != PH_DONE) {
        switch (phase) {
        // ------------------------------------------------------------------ ALM outer (K5)
        case PH_OUTER_BEGIN: {
            for (int kk = lane; kk < m; kk += 64) { // detail::project_y
                double lbd, ubd;
                constraint_bounds(c, kk % c.sm, lbd, ubd);
                const double ylo = isinf(lbd) ? 0.0 : -c.Mcap, yhi = isinf(ubd) ? 0.0 : c.Mcap;
                w.y[am + kk] = fmin(fmax(w.y[am + kk], ylo), yhi);
            }
            const int out_of_pen = (first ? init_red == c.max_num_initial_retries
                                          : pen_red == c.max_num_retries) ||
                                   (init_red + pen_red == c.max_total_num_retries);
            out_of_iter = outer + 1 == c.max_outer;
            const int budget = c.max_total_inner - inner_tot;
            max_it = c.max_iter < budget ? c.max_iter : budget;
            overwrite = out_of_iter || out_of_pen || (max_it >= budget);
            // A failed inner solve that the outer loop backtracks over WITHOUT constraints (m = 0) leaves x
            // where it was and changes only the tolerance: the retry is the same deterministic computation
            // -- same start, same L-BFGS reset, same evaluations -- and ends the same way unless the new
            // tolerance lets the stop test fire earlier (it cannot while it is below every stop measure the
            // failed run saw), the iteration limit moved inside the run, or this retry is the one whose
            // result is kept (overwrite).  alpaqa walks through up to 20 such retries, ~120 evaluations
            // each, and a batch waits for the one agent that does (the Pacejka benchmark's slowest: 2 400 of
            // its 3 081 evaluations).  The memo replays the outcome and the counts, not the evaluations: the
            // result and every statistic are those of the run that was not repeated.
            if (!c.no_memo && memo_status != 0 && m == 0 && !overwrite && c.max_total_evals == 0 &&
                memo_iters < max_it && eps < memo_mineps) {
                ps_status = memo_status; ps_iters = memo_iters; ps_eps = memo_eps;
                nevals += memo_evals;
                phase = PH_INNER_EXIT;
                break;
            }
            memo_status = 0;
            run_mineps = INFINITY; run_ev0 = nevals;
            // inner solver start: xk <- x, L-BFGS reset, Lipschitz estimate by finite differences
            lidx = 0; lfull = 0; noprog = 0; k = 0; spec = 0;
            const Row<NE> x = ldrow<NE>(w.xo + an, n, lane);
            Row<NE> xh;
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const double h = fmax(fabs(x.v[e] * c.lip_eps), c.lip_delta);
                xh.v[e] = x.v[e] + h;
                if (lane + 64 * e < n) s += h * h;
            }
            X = x;
            strow<NE>(w.xk + an, n, lane, x); strow<NE>(w.xe + an, n, lane, xh);
            hn2 = wave_sum_n(s, n);
            req = REQ_GRAD; phase = PH_W_INIT_H;
        } break;
        case PH_W_INIT_H: {
            Q = GE;                                   // grad(x + h)
            strow<NE>(w.q + an, n, lane, Q);
            strow<NE>(w.xe + an, n, lane, X);
            req = REQ_GRAD; phase = PH_W_INIT_X;
        } break;
        case PH_W_INIT_X: {
            psik = psie;
            const Row<NE> g = GE, gh = Q;
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < NE; e++) { const double dd = gh.v[e] - g.v[e]; s += dd * dd; }
            G = g;
            strow<NE>(w.gk + an, n, lane, g);
            const double dn2 = wave_sum_n(s, n);
            // std::clamp semantics: a NaN estimate stays NaN (fmin/fmax would turn it into L_min)
            const double L0 = sqrt(dn2) / sqrt(hn2);
            Lk = L0 < c.L_min ? c.L_min : (c.L_max < L0 ? c.L_max : L0);
            if (!isfinite(Lk) || psik != psik) {
                ps_status = ST_NOTFINITE; ps_iters = 0; ps_eps = INFINITY;
                phase = PH_INNER_EXIT; break;
            }
            gamma = c.Lgamma / Lk;
            tau = NAN;
            prox_to_xe<NE>(c, w.xe + an, n, lane, X, g, gamma, t_pp, t_gp); pp = t_pp; gp = t_gp;
            gamma_top = gamma;
            req = REQ_COST; phase = PH_W_DL;
        } break;
        // ------------------------------------------------------ descent lemma at the iterate (K2)
        case PH_W_DL: {
            psixh = psie;
            for (int kk = lane; kk < m; kk += 64) w.yhx[am + kk] = w.yhe[am + kk];
            const double margin = (1.0 + fabs(psik)) * c.qub_tol;
            if (psixh - psik > gp + 0.5 * Lk * pp + margin && Lk * 2.0 <= c.L_max) {
                Lk *= 2.0; gamma /= 2.0;
                prox_to_xe<NE>(c, w.xe + an, n, lane, X, G, gamma, t_pp, t_gp); pp = t_pp; gp = t_gp;
                req = REQ_COST; // stay in PH_W_DL
                break;
            }
            if (k > 0 && gamma != gamma_top) { lidx = 0; lfull = 0; }
            phik = psik + pp / (2.0 * gamma) + gp;
            phase = PH_AFTER_DL;
        } break;
        // ------------------------------------------------------------------ iteration top
        case PH_TOP: {
            gamma_top = gamma;
            if (k > 0 && c.hess_heuristic > 0 && k % c.hess_heuristic == 0) {
                // step-size heuristic (controller.py:32): FD Hessian-vector product along grad
                const Row<NE> x = X, g = G;
                double s = 0.0;
#pragma unroll
                for (int e = 0; e < NE; e++) s += x.v[e] * x.v[e];
                const double h = fd_step(wave_sum_n(s, n));
                Row<NE> xh;
#pragma unroll
                for (int e = 0; e < NE; e++) xh.v[e] = x.v[e] + h * g.v[e];
                strow<NE>(w.xe + an, n, lane, xh);
                hfd = h;
                req = REQ_GRAD; phase = PH_W_HEUR;
                break;
            }
            phase = PH_AFTER_DL;
        } break;
        case PH_W_HEUR: {
            const Row<NE> g = G, gh = GE;
            double gHg = 0.0, gg = 0.0;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const double Hv = (gh.v[e] - g.v[e]) / hfd;
                gHg += g.v[e] * Hv; gg += g.v[e] * g.v[e];
            }
            wave_sum2_n(gHg, gg, n);
            const double eta = gg / gHg;
            if (eta > 0.0 && isfinite(eta) && eta * c.Lgamma > gamma) {
                Lk = 1.0 / eta;
                gamma = c.Lgamma / Lk;
                prox_to_xe<NE>(c, w.xe + an, n, lane, X, g, gamma, t_pp, t_gp); pp = t_pp; gp = t_gp;
                req = REQ_COST; phase = PH_W_DL;
                break;
            }
            phase = PH_AFTER_DL;
        } break;
        // ------------------------------------------- stop test + structured direction (K3 setup)
        case PH_AFTER_DL: {
            const double epsk = sqrt(pp) / gamma; // ProjGradNorm2, controller.py:29
            if (epsk < run_mineps) run_mineps = epsk;
            const int stop = epsk <= eps ? ST_CONVERGED
                           : (c.max_total_evals > 0 && nevals >= c.max_total_evals) ? ST_MAXTIME
                           : k == max_it ? ST_MAXITER
                           : !isfinite(epsk) ? ST_NOTFINITE
                           : noprog > c.max_no_progress ? ST_NOPROGRESS : ST_UNKNOWN;
            if (stop != ST_UNKNOWN) {
                // (deviation, for the caller's safety: a NotFinite inner solve never hands back its iterate;
                // alpaqa would move a NaN xhat into x when always_overwrite_results is set)
                if (stop == ST_CONVERGED || (overwrite && stop != ST_NOTFINITE)) {
                    // x <- xhat, y <- yhat(xhat), err_z = g(xhat) - Pi_D(g(xhat) + y/Sigma)
                    Row<NE> x = X;
                    const Row<NE> g = G;
#pragma unroll
                    for (int e = 0; e < NE; e++) x.v[e] = x.v[e] + prox_p(c, par, x.v[e], g.v[e], gamma);
                    strow<NE>(w.xo + an, n, lane, x);
                    for (int kk = lane; kk < m; kk += 64) {
                        const double yh = w.yhx[am + kk];
                        w.e2[am + kk] = (yh - w.y[am + kk]) / w.Sig[am + kk];
                        w.y[am + kk] = yh;
                    }
                    psi_out = psixh;
                }
                ps_status = stop; ps_iters = k; ps_eps = epsk;
                phase = PH_INNER_EXIT;
                break;
            }
            nJ = 0;
            phase = PH_LS_INIT;
            const int spec_ok = spec != 0 && spec_gamma == gamma;
            spec = 0;
            if (k > 0) {
                const Row<NE> x = X, g = G;
                Row<NE> qv;
                double cntJ = 0.0, xx = 0.0;
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const bool valid = lane + 64 * e < n;
                    const bool in = in_J(c, par, x.v[e], g.v[e], gamma);
                    qv.v[e] = in ? 0.0 : prox_p(c, par, x.v[e], g.v[e], gamma);
                    if (valid) { cntJ += in ? 1.0 : 0.0; xx += x.v[e] * x.v[e]; }
                }
                wave_sum2_n(cntJ, xx, n);
                nJ = (int)cntJ;
                // (the q row is written once, by PH_LS_INIT -- or below when the step ends here: every
                // store issued before the wait for the history would be drained by it)
                if (nJ == n) {
#pragma unroll
                    for (int e = 0; e < NE; e++) qv.v[e] = -g.v[e];
                    Q = qv;
                } else {
                    Q = qv;
                    if (nJ > 0) {
                        // Hessian-vector product of the active part by finite differences
                        const double h = fd_step(xx);
                        Row<NE> xh;
#pragma unroll
                        for (int e = 0; e < NE; e++) xh.v[e] = x.v[e] + h * qv.v[e];
                        hfd = h;
                        phase = PH_W_HESS;
                        if (spec_ok) {
                            GE = in.GE2;      // grad(x + h q_J) is already there (channel 2)
                            nevals += 1; n_used = 1;
                        } else {
                            strow<NE>(w.xe + an, n, lane, xh);
                            strow<NE>(w.q + an, n, lane, qv);
                            req = REQ_GRAD;
                        }
                    }
                }
            }
        } break;
        case PH_W_HESS: {
            const Row<NE> x = X, g = G, gh = GE;
            Row<NE> qv = Q;
#pragma unroll
            for (int e = 0; e < NE; e++)
                if (in_J(c, par, x.v[e], g.v[e], gamma)) qv.v[e] = -g.v[e] - (gh.v[e] - g.v[e]) / hfd;
            Q = qv;
            phase = PH_LS_INIT;
        } break;
        // ------------------------------------------------------------------ line search (K4)
        case PH_LS_INIT: {
            Row<NE> qv = Q;
            if (k > 0 && nJ > 0) {
                const Row<NE> x = X, g = G;
                bool inj[NE];
#pragma unroll
                for (int e = 0; e < NE; e++) inj[e] = lane + 64 * e < n && in_J(c, par, x.v[e], g.v[e], gamma);
                const double *Sa = w.S + (size_t)a * c.M * n, *Ya = w.Y + (size_t)a * c.M * n;
                const double *const Sg = Sa, *const Yg = Ya;
                if (MC < 0) {
                    if (!hist_ready && (lidx | lfull) != 0) hist_dma(Sa, Ya, hist, P * n, (lfull ? c.M : (int)lidx) * n, lane);
                    hist_ready = false;
                    if (!hist_landed) hist_wait();
                    Sa = hist; Ya = hist + P * n;
                }
                const bool ok = lbfgs_two_loop<NE, MC>(c, Sa, Ya, n, lane, inj, lidx, lfull, qv, lb_rows, P, Sg, Yg);
                if (!ok) {
#pragma unroll
                    for (int e = 0; e < NE; e++) if (inj[e]) qv.v[e] *= gamma;
                }
                Q = qv;
            }
            if (k > 0) strow<NE>(w.q + an, n, lane, qv);
            tau = 1.0;
            sigpp = (1.0 - gamma * Lk) * pp / (2.0 * gamma);
            if (k == 0) tau = 0.0;
            else {
                bool fin = true;
#pragma unroll
                for (int e = 0; e < NE; e++) fin = fin && isfinite(qv.v[e]);
                if (__ballot(!fin) != 0ull) { tau = 0.0; lidx = 0; lfull = 0; }
                else if (nJ == 0) tau = 0.0;
            }
            phase = PH_LS_TRIAL;
        } break;
        case PH_LS_TRIAL: {
            Ln = Lk; gamman = gamma;
            fallback = tau / 2.0 < c.tau_min; // safe prox step: x+ = xhat, psi+ = psi(xhat)
            const Row<NE> x = trial_point<NE>(c, par, X, G, Q, gamma, tau, fallback != 0);
            XN = x;
            strow<NE>(w.xn + an, n, lane, x); strow<NE>(w.xe + an, n, lane, x);
            // (round path: the entry is marked as "gradient at a trial point" whether or not the NEXT launch will
            // carry thread-per-agent blocks -- the host decides that per launch -- see CHAIN_BIT)
            req = REQ_GRAD | (allow_chain ? REQ_CHAIN : 0); phase = PH_W_LS_G;
        } break;
        case PH_W_LS_G: {
            psin = fallback ? (double)psixh : (double)psie;
            // the gradient at x+ stays in the ge row until the next gradient evaluation
            prox_to_xe<NE>(c, w.xe + an, n, lane, XN, GE, gamman, t_pp, t_gp); ppn = t_pp; gpn = t_gp;
            req = REQ_COST; phase = PH_W_LS_C;
            speculate(gamman);
        } break;
        case PH_W_LS_C: {
            psixhn = psie;
            for (int kk = lane; kk < m; kk += 64) w.yhxn[am + kk] = w.yhe[am + kk];
            const double margin_dl = (1.0 + fabs(psin)) * c.qub_tol;
            if (psixhn - psin > gpn + 0.5 * Ln * ppn + margin_dl && Ln * 2.0 <= c.L_max) {
                Ln *= 2.0; gamman /= 2.0;
                prox_to_xe<NE>(c, w.xe + an, n, lane, XN, GE, gamman, t_pp, t_gp); ppn = t_pp; gpn = t_gp;
                req = REQ_COST; // stay
                speculate(gamman);
                break;
            }
            const double phin = psin + ppn / (2.0 * gamman) + gpn;
            const double ls_cond = phin - (phik - sigpp);
            const double margin = (1.0 + fabs(phik)) * c.qub_tol;
            tau /= 2.0;
            // (written so that a NaN condition -- a trial point whose evaluation overflowed -- counts as a
            // failed trial like +inf does, instead of being accepted because NaN compares false)
            if (!(ls_cond <= margin) && tau >= c.tau_min) {
                // (the direction is not among the rows fetched for this phase: a failed trial is the rare case)
                if (!in.has_q) Q = ldrow<NE>(w.q + an, n, lane);
                phase = PH_LS_TRIAL; spec = 0; break;
            }
            // accept x+ : L-BFGS update with (x+ - x, grad+ - grad)
            if (gamma != gamman) { lidx = 0; lfull = 0; }
            {
                const double min_div = sqrt(DBL_MIN);
                const Row<NE> x = X, xp = XN, g = G, gq = GE;
                Row<NE> s, yv;
                double ys = 0.0, ss = 0.0;
                bool same = true;
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    s.v[e] = xp.v[e] - x.v[e]; yv.v[e] = gq.v[e] - g.v[e];
                    ys += yv.v[e] * s.v[e]; ss += s.v[e] * s.v[e];
                    same = same && (xp.v[e] == x.v[e]);
                }
                wave_sum2_n(ys, ss, n);
                const bool all_same = __ballot(!same) == 0ull;
                // the pair goes into the free ring slot; it joins the history only if the
                // curvature test accepts it
                if (MC < 0 && hist_ready) {
                    // the prefetched LDS copy of the history gets the new pair too; the wait comes before
                    // this step's first global stores so that it does not have to drain them
                    hist_wait(); hist_landed = true;
                    if (lidx < P) {
#pragma unroll
                        for (int e = 0; e < NE; e++) {
                            const int j = lane + 64 * e;
                            if (j < n) { hist[(int)lidx * n + j] = s.v[e]; hist[(P + (int)lidx) * n + j] = yv.v[e]; }
                        }
                    }
                }
                strow<NE>(w.S + ((size_t)a * c.M + lidx) * n, n, lane, s);
                strow<NE>(w.Y + ((size_t)a * c.M + lidx) * n, n, lane, yv);
                X = xp; G = gq;
                strow<NE>(w.xk + an, n, lane, xp); strow<NE>(w.gk + an, n, lane, gq);
                const bool valid = isfinite(ys) && !(ss < min_div) && !(ys < min_div);
                if (valid) { lidx = lidx + 1 < c.M ? lidx + 1 : 0; lfull |= lidx == 0; }
                if (noprog > 0 || k % c.max_no_progress == 0) noprog = all_same ? noprog + 1 : 0;
            }
            for (int kk = lane; kk < m; kk += 64) w.yhx[am + kk] = w.yhxn[am + kk];
            Lk = Ln; gamma = gamman; psik = psin; psixh = psixhn; phik = phin; gp = gpn; pp = ppn;
            k++;
            phase = PH_TOP;
        } break;
        // ------------------------------------------------------------ ALM outer update (K5)
        case PH_INNER_EXIT: {
            const int conv = ps_status == ST_CONVERGED;
            inner_fail += !conv;
            inner_tot += ps_iters;
            if (ps_status == ST_NOTFINITE && ps_iters == 0) {
                // psi or its gradient is non-finite AT the point the inner solve starts from: no penalty
                // or tolerance change repairs that, the solve ends here with the inner status (a failure
                // at controller.py:64), U untouched.  alpaqa 0.0.1 would walk through its retries, two
                // evaluations each, and end as MaxIter after max_outer of them -- a 2000-round straggler
                // for the whole batch.  A NotFinite that shows up later in an inner solve (a non-finite
                // stop measure at an iterate: a line-search trial that overflowed is a FAILED trial here,
                // see PH_W_LS_C) takes the ordinary not-converged path below.
                out_eps = ps_eps; out_delta = ne1;
                status = ST_NOTFINITE;
                outer += 1;
                phase = PH_DONE;
                break;
            }
            const int out_of_time = inner_tot >= c.max_total_inner ||
                                    (c.max_total_evals > 0 && nevals >= c.max_total_evals);
            const int backtrack = !conv && !overwrite && !out_of_time;
            // (memo for PH_OUTER_BEGIN: only a run that ended without progress / non-finite inside its limits)
            if (backtrack && m == 0 && (ps_status == ST_NOPROGRESS || ps_status == ST_NOTFINITE)) {
                if (memo_status == 0) {       // a real run: remember it (a replayed one keeps the memo as it is)
                    memo_status = ps_status; memo_iters = ps_iters; memo_evals = nevals - run_ev0;
                    memo_mineps = run_mineps; memo_eps = ps_eps;
                }
            } else {
                memo_status = 0;
            }
            if (backtrack) {
                if (!first) {
                    Delta = fmax(1.0, Delta * c.Delta_lower);
                    if (HASM) update_penalty(c, w, am, lane, Delta, first, ne1);
                    rho_alm = fmin(0.5, rho_alm * c.rho_increase);
                    eps = fmax(rho_alm * eps_old, c.alm_eps);
                    pen_red += 1;
                } else {
                    for (int kk = lane; kk < m; kk += 64) w.Sig[am + kk] *= c.Sigma0_lower;
                    eps *= c.eps0_increase;
                    init_red += 1;
                }
            } else {
                double mx = 0.0;
                for (int kk = lane; kk < m; kk += 64) { // error2.swap(error1); ne1 = ||error1||_inf
                    const double t = w.e1[am + kk], e = w.e2[am + kk];
                    w.e1[am + kk] = e; w.e2[am + kk] = t;
                    const double ae = fabs(e);
                    mx = (ae > mx || ae != ae) ? ae : mx;   // NaN-propagating max
                }
                ne1 = __ballot(mx != mx) != 0ull ? NAN : wave_max(mx);
                const int alm_conv = ps_eps <= c.alm_eps && conv && ne1 <= c.alm_delta;
                if (alm_conv || out_of_iter || out_of_time) {
                    out_eps = ps_eps; out_delta = ne1;
                    status = alm_conv ? ST_CONVERGED : out_of_time ? ST_MAXTIME : ST_MAXITER;
                    outer += 1;
                    phase = PH_DONE;
                    break;
                }
                for (int kk = lane; kk < m; kk += 64) { // Sigma_old.swap(Sigma)
                    const double t = w.Sig_old[am + kk];
                    w.Sig_old[am + kk] = w.Sig[am + kk]; w.Sig[am + kk] = t;
                }
                if (HASM) update_penalty(c, w, am, lane, Delta, first, ne1);
                eps_old = eps;
                eps = fmax(rho_alm * eps, c.alm_eps);
                first = 0;
            }
            outer += 1;
            phase = outer >= c.max_outer ? PH_DONE : PH_OUTER_BEGIN;
        } break;
        default:
            phase = PH_DONE;
            break;
        }
    }
    if ((req & (REQ_GRAD | REQ_COST)) != 0) nevals += 1; // speculative evaluations count when consumed
    if ((req & (REQ_GRAD | REQ_SPEC)) != 0) n_grad = 1;

    phase.put(rv, lane); k.put(rv, lane);
    // write the record back (lane s stores slot s); the two counters accumulate over the solve
    rv += lane == R_NGRAD ? (double)n_grad : lane == R_LBROWS ? (double)lb_rows : lane == R_NSPEC ? (double)n_spec
        : lane == R_NSPEC_USED ? (double)n_used : lane == R_NCOST ? (double)((req & REQ_COST) != 0) : 0.0;
    recp[lane] = rv;
    return req;
}

// ---------------------------------------------------------------- PH_W_LS_G by one THREAD (CHAIN_BIT)
// The wave-per-agent state machine sums a vector (element j on lane j) as a balanced tree: pairs, quads,
// halves of eight and rows of sixteen lanes in lane order (row_sum16: every exchange adds commuting operands),
// then the rows as (r0 + r1) + (r2 + r3) (cross_rows; rows that hold no element are left out, which is adding
// an exact zero).  A thread that meets the elements two at a time -- stage k = elements 2k, 2k + 1 -- in
// DESCENDING stage order forms the same tree with three holders:
// same operands at every node, same bits.  Elements beyond n are the zeros the wave adds.
struct TreeSum {
    double h1 = 0.0, h2 = 0.0, h3 = 0.0, r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
    __device__ __forceinline__ void add(int k, double e0, double e1)          // k wave-uniform
    {
#pragma clang fp contract(off)
        const double t = e0 + e1;                         // lanes 2k, 2k + 1
        if (k & 1) { h1 = t; return; }
        const double q = t + h1;                          // stages k, k + 1: a quad of lanes
        h1 = 0.0;
        if (k & 2) { h2 = q; return; }
        const double hf = q + h2;                         // stages k .. k + 3: half a row
        h2 = 0.0;
        if (k & 4) { h3 = hf; return; }
        const double row = hf + h3;                       // stages k .. k + 7: a row of sixteen lanes
        h3 = 0.0;
        const int ri = k >> 3;
        if (ri == 0) r0 = row; else if (ri == 1) r1 = row; else if (ri == 2) r2 = row; else r3 = row;
    }
    __device__ __forceinline__ double total(int n) const  // cross_rows<NROWS(n)>
    {
#pragma clang fp contract(off)
        return n <= 32 ? r0 + r1 : n <= 48 ? (r0 + r1) + r2 : (r0 + r1) + (r2 + r3);
    }
};

// first lane of the wave among those that call it with on = true does one atomic for all of them; returns the
// caller's position (callers with on = false: undefined).  May be called inside divergent code.
__device__ __forceinline__ int wave_append(int *counter, bool on)
{
    const unsigned long long bal = __ballot(on);
    if (bal == 0ull) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = (int)__builtin_ctzll(bal);
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, (int)__popcll(bal));
    base = __shfl(base, leader);
    return base + (int)__popcll(bal & ((1ull << lane) - 1ull));
}

// A workgroup of the step-kernel launch (blocks nstep ..) that serves 64 gradient slots of the FINISHED round:
// for every slot whose entry carries CHAIN_BIT it does what PH_W_LS_G of advance_agent does (NE = 1) -- psi+ =
// psi (or psi(xhat) on the safe step), prox step at the trial point -> xe row, ||p||^2 and grad'p, the
// speculative Hessian-vector point -> xe2 row, the cost request (and the speculative gradient request) onto
// this round's lists, the record left as advance_agent leaves it -- with ONE THREAD per agent.  The gradient rows
// come in through an LDS tile (coalesced, all four waves), the trial point from the slot-indexed useq scratch of
// the finished round (coalesced; K1a of this round has not run yet), the results leave through the tiles again;
// one wave does the arithmetic, 40 elements per lane, with the sums formed as the wave reductions form them.
// ~2 000 wave-instructions per 64 slots, against ~350 per AGENT for the wave-per-agent step.
constexpr int CHAIN_SLOTS = 64;   // slots per workgroup: one tile of 64 x (n + 1) doubles, 21 KB at n = 40 -- under the 38 KB of
                                  // history the wave-per-agent blocks of the launch hold, so that four workgroups of either
                                  // kind share a CU
__device__ __forceinline__ void chain_block(const DevCfg &c, const Workspace &w, int cb, int gpad, int par,
                                            int *__restrict__ lists_out, int *__restrict__ counts_out, double *lds)
{
#pragma clang fp contract(off)   // fixed roundings: the same bits as the wave-per-agent PH_W_LS_G
    // ONE wave does the work of a chain block (the other three of the 256-thread workgroup leave at once and give
    // their slots back: a thread per agent fills a wave with 64 agents, and a wave that only waits at barriers would
    // hold registers the wave-per-agent blocks of the launch can use)
    if (threadIdx.x >= 64) return;
    const int n = c.n, N = c.N, ld = n + 1, t = threadIdx.x;
    const int slot0 = cb * CHAIN_SLOTS;
    // Bound guards (round 4; uniform, off the hot path): the slot count comes from a device counter and the agent ids
    // from the slot table of the finished round -- neither is trusted further than the arrays they index.  A view holds
    // at most one gradient request per agent (<= Bp slots, padded to 64) inside its 2 Bp + 64 slot interval, its lists
    // hold Bp entries each, and an agent id is < B.
    if (gpad > w.Bp + 64) gpad = w.Bp + 64;
    if (slot0 < 0 || slot0 >= gpad) return;              // uniform: no gradient slots here
    double *tA = lds;
    int *s_agent = (int *)(tA + CHAIN_SLOTS * ld);       // agent of the slot or -1
    const int total = CHAIN_SLOTS * n;
    {
        const int uslot = slot0 + t;
        const int raw = uslot < gpad ? w.agent_of[uslot] : -1;
        const bool on = raw >= 0 && (raw & CHAIN_BIT) != 0 && (raw & CH2_BIT) == 0 && (raw & AGENT_MASK) < w.B;
        s_agent[t] = on ? (raw & AGENT_MASK) : -1;
    }
    __builtin_amdgcn_wave_barrier();                     // one wave: LDS is in order
    for (int base = 0; base < total; base += 64 * 8) {   // gradient rows -> the tile (coalesced), eight loads in flight per lane
        double v[8];
        int off[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * 64 + t;
            const int r = idx / n, j = idx - r * n;
            const int ar = idx < total ? s_agent[r] : -1;
            off[u] = ar >= 0 ? r * ld + j : -1;
            v[u] = ar >= 0 ? w.ge[(size_t)ar * n + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (off[u] >= 0) tA[off[u]] = v[u];
    }
    __builtin_amdgcn_wave_barrier();
    const int a = s_agent[t];
    if (a >= 0) {
        double *r = w.rec + (size_t)a * REC;
        const double gm = r[R_GAMMAN];
        const double *useq = w.useq + slot0 + t;
        const size_t St = (size_t)w.St;
        TreeSum sxx;                                     // speculate(): xx = sum of squares of the trial point
        for (int k = N - 1; k >= 0; k--) {
            const double x0 = useq[(size_t)(2 * k) * St], x1 = useq[(size_t)(2 * k + 1) * St];
            sxx.add(k, 0.0 + x0 * x0, 0.0 + x1 * x1);
        }
        const double h = fd_step(sxx.total(n));
        double *ga = tA + t * ld;
        double *x2 = w.xe2 + (size_t)a * n;              // the speculative point goes straight to its row (16 bytes per stage)
        TreeSum spp, sgp;
        int cnt = 0;
        for (int k = N - 1; k >= 0; k--) {
            const double x0 = useq[(size_t)(2 * k) * St], x1 = useq[(size_t)(2 * k + 1) * St];
            const double g0 = ga[2 * k], g1 = ga[2 * k + 1];
            const double p0 = prox_p(c, 0, x0, g0, gm), p1 = prox_p(c, 1, x1, g1, gm);   // prox_to_xe
            ga[2 * k] = x0 + p0; ga[2 * k + 1] = x1 + p1;
            spp.add(k, fma(p0, p0, 0.0), fma(p1, p1, 0.0));
            sgp.add(k, fma(g0, p0, 0.0), fma(g1, p1, 0.0));
            if (!c.no_spec) {                                                          // speculate()
                const bool in0 = in_J(c, 0, x0, g0, gm), in1 = in_J(c, 1, x1, g1, gm);
                cnt += (in0 ? 1 : 0) + (in1 ? 1 : 0);
                const double q0 = in0 ? 0.0 : p0, q1 = in1 ? 0.0 : p1;
                x2[2 * k] = x0 + h * q0; x2[2 * k + 1] = x1 + h * q1;
            }
        }
        const bool spec = !c.no_spec && cnt > 0 && cnt < n;
        r[R_PSIN] = rec_int_of(r[R_FALLBACK]) != 0 ? r[R_PSIXH] : r[R_PSIE];
        r[R_PPN] = spp.total(n); r[R_GPN] = sgp.total(n);
        r[R_SPEC] = rec_int(spec ? 1 : 0);
        if (spec) { r[R_SPEC_GAMMA] = gm; r[R_NSPEC] += 1.0; r[R_NGRAD] += 1.0; }
        r[R_NEVALS] = rec_int(rec_int_of(r[R_NEVALS]) + 1); r[R_NCOST] += 1.0;
        r[R_PHASE] = rec_int(PH_W_LS_C + chain_tag(par));
        const int pc = wave_append(&counts_out[1], true);
        if (pc >= 0 && pc < w.Bp) lists_out[(size_t)w.Ls + pc] = a;
        const int pg = wave_append(&counts_out[0], spec);
        if (spec && pg >= 0 && pg < w.Bp) lists_out[pg] = a | CH2_BIT;
    }
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < total; base += 64 * 8) {   // the tile (now xhat+) -> xe rows
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * 64 + t;
            const int r = idx / n, j = idx - r * n;
            const int ar = idx < total ? s_agent[r] : -1;
            if (ar >= 0) w.xe[(size_t)ar * n + j] = tA[r * ld + j];
        }
    }
}

// One wave per agent at a time; a 256-thread workgroup owns 64 consecutive agents (each of its
// 4 waves walks 16 of them), collects their requests in LDS and appends them to the round's
// gradient / cost work lists with one atomic per list, in agent order.
constexpr int STEP_WAVES = 4;
#ifndef MPC_STEP_WAVES_LEAN
#define MPC_STEP_WAVES_LEAN 4
#endif

// (the unconstrained one-element-per-lane variant with the history in LDS -- the benchmark's -- needs 133
// registers: held to 128 it runs four waves per SIMD, with the LDS copy of the history capped at P pairs so that
// four workgroups share a CU; every other variant keeps its two or three)
template <int NE, int MC, bool HASM>
__global__ void __launch_bounds__(64 * STEP_WAVES, (NE == 1 && MC < 0 && !HASM) ? MPC_STEP_WAVES_LEAN : (NE == 2 && MC < 0) ? 3 : 2)
step_kernel(const DevCfg c, const Workspace w, int *__restrict__ lists_out,
            int *__restrict__ counts_out, int *__restrict__ counts_next, int apb, int nstep, int par, int P)
{
    // blocks [0, nstep): the wave-per-agent state machine; blocks beyond (c.chain): PH_W_LS_G by one thread per
    // agent for the gradient slots of the round just finished, whose count K1c left in counts_next[2] (the buffer
    // of that round: this kernel zeroes its two list counters for the round after this one, not that word)
    if ((int)blockIdx.x >= nstep) {
        extern __shared__ double s_chain[];
        if (NE == 1) chain_block(c, w, (int)blockIdx.x - nstep, counts_next[2], par, lists_out, counts_out, s_chain);
        return;
    }
    // apb = agents per workgroup (64, 16 or 4): a wave walks its agents one after the other, so a small
    // batch is spread over more workgroups (one agent per wave at apb = 4) -- latency, not throughput
    __shared__ int s_req[64];
    __shared__ int s_next;
    extern __shared__ double s_hist[];                   // MC < 0: 2 M n doubles per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#if MPC_DEV_STAMP == 3
    DevStamp stamp(blockIdx.x * STEP_WAVES + wv);
#endif
    double *hist = s_hist + (MC < 0 ? (size_t)wv * 2 * P * c.n : 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) { counts_next[0] = 0; counts_next[1] = 0; } // next round's buffer
    // Which of the workgroup's agents are still running: one coalesced look at their phase words.
    // Only the running ones are handed to the waves, so a wave never pays a memory round trip to
    // find out that an agent is finished.
    const int base = blockIdx.x * apb;
    const int phw = lane < apb && base + lane < w.B ? rec_int_of(w.rec[(size_t)(base + lane) * REC + R_PHASE]) : 0;
    // (PH_DONE == 0; with c.chain an agent that waits in PH_W_LS_G is served by a chain_block of this launch)
    const bool runnable = phw != 0 && !(c.chain && (phw == PH_W_LS_G || phw == PH_W_LS_C + chain_tag(par)));
    const unsigned long long act = __ballot(runnable);
    const int rank = __popcll(act & ((1ull << lane) - 1ull));
    const int nact = __popcll(act);
    if (threadIdx.x == 0) s_next = 0;
    if (wv == 0) s_req[lane] = REQ_NONE;
    __syncthreads();
    // the waves take the running agents from a shared counter, one ahead of the one they work on (its
    // rows are in flight meanwhile): an agent-step costs between ~0.3 and ~3 us depending on its phase,
    // and a static deal leaves three waves waiting for the unlucky one
    auto claim = [&]() -> int {
        int i = 0;
        if (lane == 0) i = atomicAdd(&s_next, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= nact) return -1;
        return (int)__builtin_ctzll(__ballot(runnable && rank == i));
    };
    AgentIn<NE> nxt;
    int loc = claim();
    // (MPC_ALL_ROWS: the six-row fetch of rounds 1 - 2, for the A/B measurement and the bit-identity test)
    const auto phase_of = [&](int l) { return c.all_rows ? -1 : (__builtin_amdgcn_readlane(phw, l) & PH_MASK); };
    if (loc >= 0) nxt = load_agent<NE>(c, w, base + loc, lane, phase_of(loc));
    while (loc >= 0) {
        const int a = base + loc;
        const AgentIn<NE> cur = nxt;
        const int loc_next = claim();
        if (loc_next >= 0) nxt = load_agent<NE>(c, w, base + loc_next, lane, phase_of(loc_next)); // in flight during agent a
        bool hist_ready = false;
        if (MC < 0) {
            // An agent that comes back from its Hessian-vector evaluation (or from the cost of a trial
            // whose speculative gradient is there) runs the two-loop almost first thing: start the
            // LDS-DMA of its history now.  Issued BEHIND the next agent's row loads: the wait for the
            // history drains the wave's vector-memory queue in order, so nothing younger than what it
            // needs should be in it.
            const int rlo = __double2loint(cur.rv);      // (the integers of the record: rec_int)
            const int ph = __builtin_amdgcn_readlane(rlo, R_PHASE) & PH_MASK;
            const int hi = __builtin_amdgcn_readlane(rlo, R_LIDX), hf = __builtin_amdgcn_readlane(rlo, R_LFULL);
            const int hl = hi | hf;
            const int sp = __builtin_amdgcn_readlane(rlo, R_SPEC);
            if ((ph == PH_W_HESS || (ph == PH_W_LS_C && sp != 0)) && hl != 0) {
                hist_dma(w.S + (size_t)a * c.M * c.n, w.Y + (size_t)a * c.M * c.n, hist, P * c.n,
                         (hf ? c.M : hi) * c.n, lane);
                hist_ready = true;
            }
        }
#if MPC_DEV_STAMP == 3
        stamp.nfall++;                                   // agent-steps of this wave
        const long long tv0 = __builtin_amdgcn_s_memrealtime();
        const int ph_in = __builtin_amdgcn_readlane(__double2loint(cur.rv), R_PHASE) & PH_MASK;
#endif
        const int req = advance_agent<NE, MC, HASM>(c, w, a, lane, cur, hist, hist_ready, true, /*allow_chain=*/true, P);
#if MPC_DEV_STAMP == 3
        {   // the longest agent-step of this wave: its length in 10 ns ticks (nmid, capped at 255) and the phase it came in with (nslow)
            const int dt = (int)(__builtin_amdgcn_s_memrealtime() - tv0);
            if (dt > stamp.nmid) { stamp.nmid = dt > 255 ? 255 : dt; stamp.nslow = ph_in; }
        }
#endif
        if (lane == 0) s_req[loc] = req;
        loc = loc_next;
    }
    __syncthreads();
    if (wv == 0) {
        const int r = s_req[lane];
#pragma unroll
        for (int kind = 0; kind < 2; kind++) { // 0: gradient list (normal or channel 2), 1: cost list
            const bool on = kind == 0 ? (r & (REQ_GRAD | REQ_SPEC)) != 0 : (r & REQ_COST) != 0;
            const unsigned long long bal = __ballot(on);
            const int cnt = __popcll(bal);
            if (cnt == 0) continue;                      // uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&counts_out[kind], cnt);
            base = __builtin_amdgcn_readfirstlane(base);
            if (on) {
                const int off = __popcll(bal & ((1ull << lane) - 1ull));
                const int flag = kind == 0 ? ((r & REQ_SPEC) ? CH2_BIT : 0) | ((r & REQ_CHAIN) ? CHAIN_BIT : 0) : 0;
                lists_out[(size_t)kind * w.Ls + base + off] = (blockIdx.x * apb + lane) | flag;
            }
        }
    }
}

// solver state initialisation for a fresh solve (ALMSolver::operator() prologue)
__global__ void init_kernel(const DevCfg c, const Workspace w)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // one thread per record slot
    const int a = (int)(i / REC), slot = (int)(i % REC);
    if (a >= w.B) return;
    double v = rec_is_int(slot) ? rec_int(0) : 0.0;
    switch (slot) {
    case R_EPS: v = c.eps0; break;
    case R_EPS_OLD: v = NAN; break;
    case R_DELTA: v = c.Delta; break;
    case R_RHO: v = c.rho; break;
    case R_NE1: v = NAN; break;
    case R_OUT_EPS: v = INFINITY; break;
    case R_OUT_DELTA: v = INFINITY; break;
    case R_FIRST: v = rec_int(1); break;
    case R_PHASE: v = rec_int(c.max_outer > 0 ? PH_OUTER_BEGIN : PH_DONE); break;
    default: break;
    }
    w.rec[i] = v;
    const size_t am = (size_t)a * c.m;
    for (int kk = slot; kk < c.m; kk += REC) {
        w.Sig[am + kk] = c.Sigma0; w.Sig_old[am + kk] = NAN; w.e1[am + kk] = NAN; w.e2[am + kk] = NAN;
    }
}

// evaluation / history-read totals of a solve: one atomic per workgroup, once per solve
__global__ void __launch_bounds__(256) totals_kernel(const Workspace w)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    double ng = 0.0, nc = 0.0, lr = 0.0, ns = 0.0, nu = 0.0, le = 0.0, lh = 0.0;
    bool unfinished = false;
    if (a < w.B) {
        const double *r = w.rec + (size_t)a * REC;
        ng = r[R_NGRAD]; nc = r[R_NCOST]; lr = r[R_LBROWS]; ns = r[R_NSPEC]; nu = r[R_NSPEC_USED];
        le = r[R_LA_EVALS]; lh = r[R_LA_HITS];
        unfinished = rec_int_of(r[R_PHASE]) != PH_DONE;
    }
    ng = wave_sum(ng); nc = wave_sum(nc); lr = wave_sum(lr); ns = wave_sum(ns); nu = wave_sum(nu);
    le = wave_sum(le); lh = wave_sum(lh);
    const unsigned long long unf = __ballot(unfinished);
    if ((threadIdx.x & 63) == 0) {
        // agents a solve left unfinished (the persistent kernel's trip guard ran out): the host reports MPC_E_LIMIT
        if (unf != 0ull) atomicAdd(&w.totals[6], (unsigned long long)__popcll(unf));
        atomicAdd(&w.totals[0], (unsigned long long)ng);   // gradient evaluations executed (incl. speculative)
        atomicAdd(&w.totals[1], (unsigned long long)nc);   // cost evaluations
        atomicAdd(&w.totals[2], (unsigned long long)lr);
        atomicAdd(&w.totals[4], (unsigned long long)ns);   // speculative gradients issued / consumed
        atomicAdd(&w.totals[5], (unsigned long long)nu);
        if (le != 0.0) atomicAdd(&w.totals[7], (unsigned long long)le);   // lookahead: candidate evaluations executed
        if (lh != 0.0) atomicAdd(&w.totals[8], (unsigned long long)lh);   // ... requests served from them
    }
}

__global__ void stats_kernel(const Workspace w, double *__restrict__ stats)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= w.B) return;
    const double *r = w.rec + (size_t)a * REC;
    double *s = stats + (size_t)a * 8;
    s[0] = (double)rec_int_of(r[R_STATUS]); s[1] = (double)rec_int_of(r[R_OUTER]);
    s[2] = (double)rec_int_of(r[R_INNER_TOT]); s[3] = (double)rec_int_of(r[R_INNER_FAIL]);
    s[4] = r[R_OUT_EPS]; s[5] = r[R_OUT_DELTA]; s[6] = r[R_PSI_OUT]; s[7] = (double)rec_int_of(r[R_NEVALS]);
}

} // namespace mpc
