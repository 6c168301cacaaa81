// mpc_eval.hpp -- K1: cost (+ gradient) of the control sequences of the agents on the work lists.
//
// The horizon is a serial recurrence, but only the state rollout is: everything else about an
// evaluation is independent per stage.  K1 is therefore three launches:
//   K1a rollout_kernel  one thread per agent: x_0 .. x_N (16 N dependent RHS evaluations), inputs
//                       staged through LDS (agent-major rows -> one coalesced access per agent)
//   K1b stage_kernel    one thread per (agent, stage): nearest centerline point, stage cost, ALM
//                       terms and -- for gradient requests -- the stage's cost gradient and its
//                       transition sensitivities (forward-mode tangents through the 4 RK4 steps)
//   K1c adjoint_kernel  one thread per agent: psi = sum of stage costs, and the short adjoint
//                       recursion lambda_k = A_k' lambda_{k+1} + dL_k/dx over the stored blocks
// A lone wave per SIMD issues fp64 at about half rate and, late in a solve, only a fraction of the
// agents is still active: K1b exposes N times more threads, which is where the chip gets filled.
// All scratch between the three launches is slot-indexed SoA (coalesced), L2/MALL resident.
#pragma once
#include "mpc_solver.hpp"

// occupancy the Pacejka kernels are compiled for (waves per SIMD); the values below are the measured best
// (profiles/r03_experiments.txt 15: stage kernel at two 694 -> 730 ms per solve, four-lane rollout at three 694 -> 696)
#ifndef MPC_K1B_WAVES_PAC
#define MPC_K1B_WAVES_PAC 1
#endif
#ifndef MPC_QUAD_WAVES
#define MPC_QUAD_WAVES 2
#endif

namespace mpc {

// unified slot space of a round: gradient requests [0, nG), cost requests [gpad, gpad + nC),
// gpad = nG rounded up to a multiple of 64 so that no wave mixes the two kinds
struct SlotMap {
    int nG, nC, gpad, nblk_g, nblk;
    __device__ SlotMap(const int *counts, int nG_imm, int nC_imm)
    {
        nG = counts ? counts[0] : nG_imm;
        nC = counts ? counts[1] : nC_imm;
        gpad = (nG + 63) & ~63;
        nblk_g = gpad >> 6;
        nblk = nblk_g + ((nC + 63) >> 6);
    }
};

template <int MODEL>
__global__ void __launch_bounds__(64, 2)
rollout_kernel(const DevCfg c, const Workspace w, const int *__restrict__ lists,
               const int *__restrict__ counts, int nG_imm, int nC_imm)
{
    constexpr int NX = ModelDim<MODEL>::NX;
    extern __shared__ double lds[];
    double *tile = lds;
    int *s_agent = (int *)(lds + 64 * (c.n + 1));
    const SlotMap sm(counts, nG_imm, nC_imm);
    const int sb = blockIdx.x;
    if (sb >= sm.nblk) return;
    const int lane = threadIdx.x;
    const bool is_g = sb < sm.nblk_g;
    const int uslot = sb * 64 + lane;
    const int kslot = is_g ? uslot : uslot - sm.gpad;              // position on its own list
    const bool active = kslot < (is_g ? sm.nG : sm.nC);
    const int *list = counts ? (is_g ? lists : lists + w.Ls) : nullptr;
    const int raw = active ? (list ? list[kslot] : kslot) : -1;  // agent id | CH2_BIT
    const int a = raw < 0 ? -1 : raw & AGENT_MASK;
    const int n = c.n, N = c.N, ld = n + 1;
    s_agent[lane] = raw;
    w.agent_of[uslot] = raw;
    __syncthreads();
    // stage in: the 64 agent-major rows of this workgroup go through LDS.  Element idx of the
    // flattened [64][n] tile belongs to row idx / n; eight independent loads are in flight per lane
    // before the first LDS write (a row-at-a-time loop would pay one memory latency per row).
    const int total = 64 * n;
    for (int base = 0; base < total; base += 64 * 8) {
        double v[8];
        int off[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * 64 + lane;
            const int r = idx / n, j = idx - r * n;
            const int ar = idx < total ? s_agent[r] : -1;
            off[u] = ar >= 0 ? r * ld + j : -1;
            v[u] = ar >= 0 ? ((ar & CH2_BIT) ? w.xe2 : w.xe)[(size_t)(ar & AGENT_MASK) * n + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (off[u] >= 0) tile[off[u]] = v[u];
    }
    __syncthreads();
    if (!active) return;
    const size_t St = (size_t)w.St; // scratch stride
    const double *urow = tile + lane * ld;
    double x[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) { x[i] = w.x0[(size_t)a * NX + i]; w.trajx[(size_t)i * St + uslot] = x[i]; }
    for (int k = 0; k < N; k++) {
        const double d = urow[2 * k], dl = urow[2 * k + 1];
        w.useq[(size_t)(2 * k) * St + uslot] = d;
        w.useq[(size_t)(2 * k + 1) * St + uslot] = dl;
        StageInput<MODEL> u;
        prep_input(c, d, dl, u);
        stage_forward<MODEL>(c, u, x);
#pragma unroll
        for (int i = 0; i < NX; i++) w.trajx[(size_t)((k + 1) * NX + i) * St + uslot] = x[i];
    }
}

// K1a of the kinematic model (nfe = 4) by TWO lanes per request: both walk the cheap heading/speed
// recursion of a stage (4 x 17 dependent fma), lane 0 evaluates the position increments -- where the
// sin/cos are -- of RK4 steps 0 and 1, lane 1 those of steps 2 and 3; they swap them and both add the
// four in step order, so the pair keeps identical copies of the state.  Twice the waves at ~0.57 of
// the chain length: a round with all agents active gives the thread-per-request kernel 1.4 waves per
// SIMD (its makespan is that of the SIMDs holding two), a sub-batch group a third of that.  The same
// helper functions with fixed roundings as rollout_kernel<KIN> and the wave-per-request kernel: same bits.
#ifndef MPC_K1A_WAVES
#define MPC_K1A_WAVES 3
#endif
template <class Put>
__device__ __forceinline__ void kin_wide_rollout(const DevCfg &c, const double (&x0)[4], double d, double dl, int lane, Put put);
__global__ void __launch_bounds__(64, MPC_K1A_WAVES)
rollout_pair_kernel(const DevCfg c, const Workspace w, const int *__restrict__ lists,
                    const int *__restrict__ counts, int nG_imm, int nC_imm)
{
#pragma clang fp contract(off)
#if MPC_DEV_STAMP == 1
    DevStamp stamp(blockIdx.x);
#endif
    constexpr int RPB = 32;                              // requests per block
    extern __shared__ double lds[];                      // [RPB][n + 1] control rows
    const SlotMap sm(counts, nG_imm, nC_imm);
    const int lane = threadIdx.x, q = lane >> 1, half = lane & 1;
    const int uslot = blockIdx.x * RPB + q;
    const int nslots = sm.nblk * 64;
    const bool is_g = uslot < sm.gpad;
    const int kslot = is_g ? uslot : uslot - sm.gpad;
    const bool active = uslot < nslots && kslot < (is_g ? sm.nG : sm.nC);
    const int *list = counts ? (is_g ? lists : lists + w.Ls) : nullptr;
    const int raw = active ? (list ? list[kslot] : kslot) : -1;
    const int n = c.n, N = c.N, ld = n + 1;
    if (half == 0 && uslot < nslots) w.agent_of[uslot] = raw;
    if (__ballot(active) == 0ull) return;                // uniform: nothing in this block
    {
        const double *__restrict__ row = ((raw & CH2_BIT) ? w.xe2 : w.xe) + (size_t)(raw & AGENT_MASK) * n;
        // eight loads in flight per lane before the first LDS write (one at a time = one memory round trip each);
        // pairs without a request (the last block of a list) keep a row of zeros and walk along: all 64 lanes
        // stay in the wave for the requests it redoes below
        for (int j0 = half; j0 < n; j0 += 16) {
            double v[8];
#pragma unroll
            for (int t = 0; t < 8; t++) v[t] = active && j0 + 2 * t < n ? row[j0 + 2 * t] : 0.0;
#pragma unroll
            for (int t = 0; t < 8; t++) if (j0 + 2 * t < n) lds[q * ld + j0 + 2 * t] = v[t];
        }
    }
    __builtin_amdgcn_wave_barrier();                     // one wave per workgroup: LDS is in order
    const int a = active ? raw & AGENT_MASK : 0;
    const size_t St = (size_t)w.St;
    const double *urow = lds + q * ld;
    double x[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = active ? w.x0[(size_t)a * 4 + i] : 0.0;
    // A stage outside the fast range (kin4_in_range: line-search trial points far outside the box, a few
    // requests in ten thousand) is not computed here: the pair stops storing, walks on with whatever the
    // straight-line code makes of it, and the wave redoes the request afterwards over all of its lanes.  (The
    // thread-per-request code for such a stage, run by the whole wave on the spot, made that wave three times as
    // long as the others: 1 - 2 % of the waves set the length of every mid-solve launch, profiles/r03_experiments.txt 27.)
    bool redo = false;
    // lane `half` stores components 2 half and 2 half + 1 of every state
    auto put = [&](int k) {
        if (!active || redo) return;
        w.trajx[(size_t)(k * 4 + 2 * half) * St + uslot] = half ? x[2] : x[0];
        w.trajx[(size_t)(k * 4 + 2 * half + 1) * St + uslot] = half ? x[3] : x[1];
    };
    put(0);
    // the stage inputs (beta, sin(beta) / lr, ...: a division chain of ~100 instructions) are prepared
    // two stages at a time, stage k + half by lane `half`, and handed to the partner by DPP
    auto stage = [&](int k, const StageInput<KIN> &u) {
        const double d = urow[2 * k], dl = urow[2 * k + 1];
        if (active) w.useq[(size_t)(2 * k + half) * St + uslot] = half ? dl : d;
        redo = redo || !kin4_in_range(c, u, x);          // (both lanes of a pair hold the same state)
#if MPC_DEV_STAMP == 1
        if (__ballot(redo) != 0ull) stamp.nfall++;
#endif
        // heading / speed at the start of the four RK4 steps, and the stage values of each
        double ph = x[2], v = x[3];
        KinRK k0, k1, k2, k3;
        const double ph0 = ph; kin_rk(c, u, v, k0); kin_next(c, k0, ph, v);
        const double ph1 = ph; kin_rk(c, u, v, k1); kin_next(c, k1, ph, v);
        const double ph2 = ph; kin_rk(c, u, v, k2); kin_next(c, k2, ph, v);
        const double ph3 = ph; kin_rk(c, u, v, k3); kin_next(c, k3, ph, v);
        // this lane's two increments: steps 2 half and 2 half + 1
        KinRK ka, kb;
        ka.v1 = half ? k2.v1 : k0.v1; ka.v2 = half ? k2.v2 : k0.v2; ka.v3 = half ? k2.v3 : k0.v3; ka.v4 = half ? k2.v4 : k0.v4;
        ka.kp1 = half ? k2.kp1 : k0.kp1; ka.kp2 = half ? k2.kp2 : k0.kp2; ka.kp3 = half ? k2.kp3 : k0.kp3;
        kb.v1 = half ? k3.v1 : k1.v1; kb.v2 = half ? k3.v2 : k1.v2; kb.v3 = half ? k3.v3 : k1.v3; kb.v4 = half ? k3.v4 : k1.v4;
        kb.kp1 = half ? k3.kp1 : k1.kp1; kb.kp2 = half ? k3.kp2 : k1.kp2; kb.kp3 = half ? k3.kp3 : k1.kp3;
        double dxa, dya, dxb, dyb;
        kin_increment(c, u, half ? ph2 : ph0, ka, !redo, dxa, dya);
        kin_increment(c, u, half ? ph3 : ph1, kb, !redo, dxb, dyb);
        const double oxa = dpp_xchg<0xB1>(dxa), oya = dpp_xchg<0xB1>(dya);   // the partner's (quad_perm [1,0,3,2])
        const double oxb = dpp_xchg<0xB1>(dxb), oyb = dpp_xchg<0xB1>(dyb);
        // the position is lane 0's to keep (it stores x and y, lane 1 heading and speed, which never see
        // the position): its own increments are steps 0 and 1, its partner's steps 2 and 3 -- lane 1
        // adds the same operands in an order that means nothing and never uses the result
        double px = x[0], py = x[1];
        px = px + dxa; py = py + dya;                                         // step 0
        px = px + dxb; py = py + dyb;                                         // step 1
        px = px + oxa; py = py + oya;                                         // step 2
        px = px + oxb; py = py + oyb;                                         // step 3
        x[0] = px; x[1] = py; x[2] = ph; x[3] = v;
        put(k + 1);
    };
    for (int k = 0; k < N; k += 2) {
        const int km = min(k + half, N - 1);
        StageInput<KIN> um, up;
#if MPC_DEV_STAMP == 1
        if (__ballot(!(fabs(urow[2 * km + 1]) <= 0.75)) != 0ull) stamp.nmid++;
        if (__ballot(!(fabs(urow[2 * km + 1]) < 1.0e5)) != 0ull) stamp.nslow++;
#endif
        prep_input(c, urow[2 * km], urow[2 * km + 1], um);
        up.ad = dpp_xchg<0xB1>(um.ad); up.beta = dpp_xchg<0xB1>(um.beta);
        up.sb_lr = dpp_xchg<0xB1>(um.sb_lr); up.cb_lr = dpp_xchg<0xB1>(um.cb_lr);
        up.dbeta = dpp_xchg<0xB1>(um.dbeta); up.mk0 = dpp_xchg<0xB1>(um.mk0); up.mk1 = dpp_xchg<0xB1>(um.mk1);
        StageInput<KIN> ua, ub;
        ua.ad = half ? up.ad : um.ad; ua.beta = half ? up.beta : um.beta; ua.sb_lr = half ? up.sb_lr : um.sb_lr;
        ua.cb_lr = half ? up.cb_lr : um.cb_lr; ua.dbeta = half ? up.dbeta : um.dbeta;
        ua.mk0 = half ? up.mk0 : um.mk0; ua.mk1 = half ? up.mk1 : um.mk1;
        ub.ad = half ? um.ad : up.ad; ub.beta = half ? um.beta : up.beta; ub.sb_lr = half ? um.sb_lr : up.sb_lr;
        ub.cb_lr = half ? um.cb_lr : up.cb_lr; ub.dbeta = half ? um.dbeta : up.dbeta;
        ub.mk0 = half ? um.mk0 : up.mk0; ub.mk1 = half ? um.mk1 : up.mk1;
        stage(k, ua);
        if (k + 1 < N) stage(k + 1, ub);
    }
    // the requests that met a stage outside the fast range, one after the other over the whole wave: per stage
    // the same choice between the two forms of the RK4 step, and the same bits, as every other K1a kernel
    unsigned long long todo = __ballot(redo && active && half == 0);
    while (todo != 0ull) {                               // uniform
        const int l0 = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int aq = __builtin_amdgcn_readlane(raw, l0) & AGENT_MASK;
        const int uq = blockIdx.x * RPB + (l0 >> 1);
        const double *uro = lds + (l0 >> 1) * ld;
        const bool stage_lane = lane < N;
        const double dq = stage_lane ? uro[2 * lane] : 0.0, dlq = stage_lane ? uro[2 * lane + 1] : 0.0;
        double xq[4];
#pragma unroll
        for (int i = 0; i < 4; i++) xq[i] = w.x0[(size_t)aq * 4 + i];
        double *const tj = w.trajx + uq;
        kin_wide_rollout(c, xq, dq, dlq, lane, [=](int k, int i, double v) { tj[(size_t)(k * 4 + i) * St] = v; });
    }
}


// K1a of the Pacejka model by FOUR lanes per request (rhs_quad): a workgroup of 64 threads takes 16
// requests.  The rollout is one serial chain of 16 N RHS evaluations whatever the batch, and a batch of
// 65 536 agents in three groups gives the thread-per-request kernel half a wave per SIMD: four times
// the waves at 0.56 of the chain length fill the chip where it was idle, and shorten the chain where
// a tail of few agents waits for it.  Same bits as rollout_kernel<PAC>.
__global__ void __launch_bounds__(64, MPC_QUAD_WAVES)
rollout_quad_kernel(const DevCfg c, const Workspace w, const int *__restrict__ lists,
                    const int *__restrict__ counts, int nG_imm, int nC_imm)
{
    constexpr int NX = 6, RPB = 16;                      // requests per block
    extern __shared__ double lds[];                      // [RPB][n + 1] control rows
#if MPC_DEV_STAMP == 4
    DevStamp stamp(blockIdx.x);
#endif
    const SlotMap sm(counts, nG_imm, nC_imm);
    const int lane = threadIdx.x, q = lane >> 2, role = lane & 3;
    const int uslot = blockIdx.x * RPB + q;
    const int nslots = sm.nblk * 64;
    const bool is_g = uslot < sm.gpad;
    const int kslot = is_g ? uslot : uslot - sm.gpad;
    const bool active = uslot < nslots && kslot < (is_g ? sm.nG : sm.nC);
    const int *list = counts ? (is_g ? lists : lists + w.Ls) : nullptr;
    const int raw = active ? (list ? list[kslot] : kslot) : -1;
    const int n = c.n, N = c.N, ld = n + 1;
    if (role == 0 && uslot < nslots) w.agent_of[uslot] = raw;
    // stage in: the quad's own row, four elements per trip
    if (active) {
        const double *__restrict__ row = ((raw & CH2_BIT) ? w.xe2 : w.xe) + (size_t)(raw & AGENT_MASK) * n;
        for (int j0 = role; j0 < n; j0 += 32) {          // eight loads in flight per lane before the first LDS write
            double v[8];
#pragma unroll
            for (int t = 0; t < 8; t++) v[t] = j0 + 4 * t < n ? row[j0 + 4 * t] : 0.0;
#pragma unroll
            for (int t = 0; t < 8; t++) if (j0 + 4 * t < n) lds[q * ld + j0 + 4 * t] = v[t];
        }
    }
    __builtin_amdgcn_wave_barrier();                     // one wave per workgroup: LDS is in order
    if (!active) return;                                 // whole quads leave together
    const int a = raw & AGENT_MASK;
    const size_t St = (size_t)w.St;
    const double *urow = lds + q * ld;
    double x[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = w.x0[(size_t)a * NX + i];
    // lane `role` stores components role and role + 4 of every state
    auto put = [&](int k) {
        w.trajx[(size_t)(k * NX + role) * St + uslot] = role == 0 ? x[0] : role == 1 ? x[1] : role == 2 ? x[2] : x[3];
        if (role < 2) w.trajx[(size_t)(k * NX + 4 + role) * St + uslot] = role == 0 ? x[4] : x[5];
    };
    put(0);
    for (int k = 0; k < N; k++) {
        const double d = urow[2 * k], dl = urow[2 * k + 1];
        if (role == 0) w.useq[(size_t)(2 * k) * St + uslot] = d;
        if (role == 1) w.useq[(size_t)(2 * k + 1) * St + uslot] = dl;
        StageInput<PAC> u;
        prep_input(c, d, dl, u);
#if MPC_DEV_STAMP == 4
        {   // stages that START from a state with a non-finite component / from a finite one outside the fast ranges
            bool fin = true;
#pragma unroll
            for (int i = 0; i < NX; i++) fin = fin && fabs(x[i]) < 1.0e300;
            const bool inr = fabs(x[2]) < 1.0e5 && fabs(dl) < 1.0e5 && (x[3] != 0.0 || (x[5] * c.lf + x[4] != 0.0 && x[5] * c.lr - x[4] != 0.0));
            if (__ballot(!fin) != 0ull) stamp.nfall++;
            if (__ballot(fin && !inr) != 0ull) stamp.nmid++;
            if (__ballot(!fin) != 0ull && k == 0) stamp.nslow++;
        }
#endif
        stage_forward_quad(c, u, x, role);
        put(k + 1);
    }
}

// K1a for few requests (late rounds, small batches): ONE WAVE per request instead of one thread.
// The thread-per-agent rollout is a serial chain of 16 N sin/cos evaluations (~43 us whatever the
// batch); here only the cheap linear heading/speed recursion stays serial (phase A), the 4 N position
// increments -- where the sin/cos are -- are computed one per lane (phase B), and the positions are
// summed in order (phase C).  Same helper functions with fixed roundings as the thread-per-agent
// kernel, so a request gets the same bits whichever kernel serves it.  Kinematic model, nfe = 4,
// N <= 64 (up to four passes of 64 RK4 steps).
// The body of the wave-per-request rollout, shared by rollout_wide_kernel (states to the slot-indexed
// scratch) and the persistent solo kernel (states to the wave's LDS): `put(k, i, v)` stores component
// i of the state at the end of stage k - 1 (k = 0: the initial state).  d / dl are the inputs of
// stage `lane` (lanes >= N: zeros).  All 64 lanes must be active.
template <class Put>
__device__ __forceinline__ void kin_wide_rollout(const DevCfg &c, const double (&x0)[4], double d, double dl, int lane, Put put)
{
    const int N = c.N;
    const bool stage_lane = lane < N;
    StageInput<KIN> u;
    prep_input(c, d, dl, u);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) put(0, i, x0[i]);
    }
    // ---- phase A: heading and speed along the horizon (uniform, serial); lane s & 63 keeps the
    // (heading, speed) at the start of RK4 step s, lane k the state at the end of stage k.  A stage outside
    // the fast range (kin4_in_range: a trial point of the line search far outside the box, as a rule) runs
    // the same recursion with the roundings of stage_forward_steps -- what the thread-per-request kernel
    // computes for it -- and its steps are marked (gen) for phase B.
    double ph = x0[2], v = x0[3];
    constexpr int PASS = 4;
    double cph[PASS] = {0.0, 0.0, 0.0, 0.0}, cv[PASS] = {0.0, 0.0, 0.0, 0.0}, eph = 0.0, ev = 0.0;
    int gen = 0;                                         // bit p: this lane's step of pass p belongs to such a stage
    bool anygen = false;                                 // uniform
#pragma unroll
    for (int p = 0; p < PASS; p++) {                     // pass p: stages 16 p .. 16 p + 15 = RK4 steps 64 p .. 64 p + 63
        const int k1 = N < 16 * (p + 1) ? N : 16 * (p + 1);
        for (int k = 16 * p; k < k1; k++) {
            StageInput<KIN> uk;
            uk.ad = rdlane(u.ad, k); uk.beta = rdlane(u.beta, k); uk.sb_lr = rdlane(u.sb_lr, k);
            const double xs[4] = {0.0, 0.0, ph, v};
            const bool inr = __ballot(!kin4_in_range(c, uk, xs)) == 0ull;   // (every lane holds the same values)
            anygen = anygen || !inr;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (lane == ((4 * k + j) & 63)) { cph[p] = ph; cv[p] = v; gen |= inr ? 0 : 1 << p; }
                KinRK kr;
                if (inr) { kin_rk(c, uk, v, kr); kin_next(c, kr, ph, v); }
                else { kin_gen_rk(c, uk, v, kr); kin_gen_next(c, kr, ph, v); }
            }
            if (lane == k) { eph = ph; ev = v; }
        }
    }
    // ---- phase B: position increment of RK4 step s = lane (and lane + 64)
    double dx[PASS] = {0.0, 0.0, 0.0, 0.0}, dy[PASS] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < PASS; p++) {
        const int s = 64 * p + lane;
        if (64 * p >= 4 * N) break;                      // uniform
        const int k = (s >> 2) < N ? (s >> 2) : N - 1;   // lanes past the horizon compute a copy, unused
        StageInput<KIN> uk;
        uk.ad = __shfl(u.ad, k); uk.beta = __shfl(u.beta, k); uk.sb_lr = __shfl(u.sb_lr, k);
        const bool g = ((gen >> p) & 1) != 0 && (s >> 2) < N;
        KinRK kr;
        kin_rk(c, uk, cv[p], kr);
        kin_increment(c, uk, cph[p], kr, !g, dx[p], dy[p]);
        if (anygen) {                                    // uniform; rare
            double gx, gy;
            kin_gen_rk(c, uk, g ? cv[p] : 0.0, kr);
            kin_gen_increment(c, uk, g ? cph[p] : 0.0, g ? cv[p] : 0.0, kr, gx, gy);
            dx[p] = g ? gx : dx[p]; dy[p] = g ? gy : dy[p];
        }
    }
    // ---- phase C: positions, summed in step order; lane k keeps the position at the end of stage k
    double px = x0[0], py = x0[1], epx = 0.0, epy = 0.0;
#pragma unroll
    for (int p = 0; p < PASS; p++) {
        const int k0 = 16 * p, k1 = N < 16 * (p + 1) ? N : 16 * (p + 1);
        for (int k = k0; k < k1; k++) {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                px = px + rdlane(dx[p], (4 * k + j) & 63);
                py = py + rdlane(dy[p], (4 * k + j) & 63);
            }
            if (lane == k) { epx = px; epy = py; }
        }
    }
    if (stage_lane) { put(lane + 1, 0, epx); put(lane + 1, 1, epy); put(lane + 1, 2, eph); put(lane + 1, 3, ev); }
}

__global__ void __launch_bounds__(256)
rollout_wide_kernel(const DevCfg c, const Workspace w, const int *__restrict__ lists,
                    const int *__restrict__ counts)
{
    const SlotMap sm(counts, 0, 0);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int uslot = blockIdx.x * 4 + wv;
    if (uslot >= sm.nblk * 64) return;
    const bool is_g = uslot < sm.gpad;
    const int kslot = is_g ? uslot : uslot - sm.gpad;
    const bool active = kslot < (is_g ? sm.nG : sm.nC);
    const int raw = active ? (is_g ? lists : lists + w.Ls)[kslot] : -1;
    if (lane == 0) w.agent_of[uslot] = raw;
    if (!active) return;
    const int a = raw & AGENT_MASK;
    const int N = c.N, n = c.n;
    const size_t St = (size_t)w.St;
    const double *__restrict__ row = ((raw & CH2_BIT) ? w.xe2 : w.xe) + (size_t)a * n;
    // lane k < N owns stage k: its inputs, and later the state at the end of the stage
    const bool stage_lane = lane < N;
    const double d = stage_lane ? row[2 * lane] : 0.0, dl = stage_lane ? row[2 * lane + 1] : 0.0;
    if (stage_lane) {
        w.useq[(size_t)(2 * lane) * St + uslot] = d;
        w.useq[(size_t)(2 * lane + 1) * St + uslot] = dl;
    }
    double x0[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x0[i] = w.x0[(size_t)a * 4 + i];
    double *const tj = w.trajx + uslot;
    kin_wide_rollout(c, x0, d, dl, lane,
                     [=](int k, int i, double v) { tj[(size_t)(k * 4 + i) * St] = v; });
}

// per (slot, stage) record written for gradient requests: dL/dx (NX), dL/du (2), T (NX x NX)
template <int MODEL> struct JacRec { static constexpr int SIZE = ModelDim<MODEL>::NX * (ModelDim<MODEL>::NX + 1) + 2; };

// the nearest centerline point of (px, py) and its two neighbours.  row: the centerline row of clp (the
// pruned searches keep tables per row) -- the same index whichever search runs
__device__ __forceinline__ void stage_geom(const DevCfg &c, const Workspace &w, const double *__restrict__ clp, int row,
                                           double px, double py, Geom &g)
{
    load_geom(c, clp, nearest_lookup(c, clp, w.near, row, px, py), g);
}

// K1b for one (request, stage): [nearest point: stage_geom], stage cost, ALM terms and -- for gradient requests --
// the stage's cost gradient and transition sensitivities.  `put(f, v)` stores field f of the stage
// record: dL/dx (NX), dL/du (2), and field JS = the stage cost (T (NX x NX): stage_sens_record).  One body for the
// two-kernel path (records in the slot-indexed scratch), the fused kernel and the persistent solo kernel
// (records in LDS).  The cost arithmetic is written with fixed roundings (no contraction, explicit
// fma): the same request must give the same bits whichever kernel this is inlined into.
template <int MODEL, class Put>
__device__ __forceinline__ void stage_record(const DevCfg &c, const Workspace &w, int a, bool ch2, bool is_g,
                                             int k, const double (&xs)[ModelDim<MODEL>::NX],
                                             const double (&xe)[ModelDim<MODEL>::NX], double d, double dl,
                                             const Geom &g, Put put)
{
    constexpr int NX = ModelDim<MODEL>::NX, JS = JacRec<MODEL>::SIZE;
    double xb[NX], ub[2] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NX; i++) xb[i] = 0.0;
    double L = is_g ? stage_cost<MODEL, true>(c, g, xe, d, dl, xb, ub)
                    : stage_cost<MODEL, false>(c, g, xe, d, dl, xb, ub);
    if (c.sm) {
#pragma clang fp contract(off)
        const size_t am = (size_t)a * c.m;
#pragma unroll
        for (int i = 0; i < NX; i++) {
            if (i < c.sm) {
                const size_t kk = am + (size_t)(k * c.sm + i);
                const double gv = stage_constraint<MODEL>(c, g, xe, i);
                double lb, ubd;
                constraint_bounds(c, i, lb, ubd);
                const double sg = w.Sig[kk];
                const double zeta = gv + w.y[kk] / sg;
                const double zhat = fmax(lb, fmin(zeta, ubd));
                const double dd = zeta - zhat;
                const double yh = sg * dd;
                L = fma(0.5 * dd, yh, L);
                if (!ch2) w.yhe[kk] = yh;
                if (is_g) stage_constraint_adjoint<MODEL>(c, g, xe, i, yh, xb);
            }
        }
    }
    put(JS, L);
    if (is_g) {
#pragma unroll
        for (int i = 0; i < NX; i++) put(i, xb[i]);
        put(NX, ub[0]);
        put(NX + 1, ub[1]);
    }
}

// Entries of the sensitivity block that are the same constant for every stage and request: the heading
// direction's row is [-dy, dx, 1, 0, ..] (stage_tangents), and the kinematic speed does not see the
// steering.  They are neither stored by K1b nor read by K1c (3 of the 22 record fields of the kinematic
// model, 4 of 44 of the Pacejka model; K1c is bound by that traffic); the recursion multiplies by the
// literal instead -- same bits as multiplying by the stored 1.0 / 0.0.
template <int MODEL> __device__ __forceinline__ constexpr bool sens_is_const(int dd, int i)
{
    return (dd == 0 && i >= 2) || (MODEL == KIN && dd == 3 && i == 3);
}
__device__ __forceinline__ constexpr double sens_const(int dd, int i) { return dd == 0 && i == 2 ? 1.0 : 0.0; }

// the transition sensitivities of a gradient request's stage (fields NX + 2 .. JS - 1 of its record).
// Called BEFORE the cost part: the sixteen values leave for memory while the nearest-point search and
// the cost are computed, and neither part holds the other's registers.
template <int MODEL, class Put>
__device__ __forceinline__ void stage_sens_record(const DevCfg &c, const double (&xs)[ModelDim<MODEL>::NX],
                                                  const double (&xe)[ModelDim<MODEL>::NX], double d, double dl, Put put)
{
    constexpr int NX = ModelDim<MODEL>::NX;
    StageInput<MODEL> u;
    prep_input(c, d, dl, u);
    double T[NX][NX];
    if constexpr (MODEL == PAC) {
        // a stage that starts from a lost state or input (pac_stage_is_lost: the rollout blew up earlier in the horizon): every
        // partial derivative there is NaN and so is every sensitivity that is computed at all; the lanes walk a
        // harmless state instead of dragging their wave through the library route 16 times, and store the NaNs
        const bool lost = pac_stage_is_lost(u, xs);
        double xp[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) xp[i] = xs[i];
        if (lost) pac_park(u, xp);
        stage_tangents_at<MODEL>(c, u, xp, xe, T);
#pragma unroll
        for (int dd = 0; dd < NX; dd++) {
#pragma unroll
            for (int i = 0; i < NX; i++)
                if (!sens_is_const<MODEL>(dd, i)) put(NX + 2 + dd * NX + i, lost ? __builtin_nan("") : T[dd][i]);
        }
        return;
    }
    stage_tangents_at<MODEL>(c, u, xs, xe, T);
#pragma unroll
    for (int dd = 0; dd < NX; dd++) {
#pragma unroll
        for (int i = 0; i < NX; i++) if (!sens_is_const<MODEL>(dd, i)) put(NX + 2 + dd * NX + i, T[dd][i]);
    }
}

// K1c for one request: psi = sum of stage costs (stage order, as main.py:36-40) and the adjoint
// recursion over the stage records; `get(k, f)` reads field f of stage k.
// (the same with the destinations given: psi -> *psi_out (may be null), gradient -> grow[2N]; the persistent kernel's
// lookahead sends candidate evaluations to its cache)
template <int MODEL, class Get>
__device__ __forceinline__ void adjoint_rec_to(const DevCfg &c, bool is_g, Get get, double *psi_out, double *grow)
{
    constexpr int NX = ModelDim<MODEL>::NX, NZ = NX - 2, JS = JacRec<MODEL>::SIZE;
    const int N = c.N;
    double psi = 0.0;
    for (int k = 0; k < N; k++) psi += get(k, JS);
    if (psi_out) *psi_out = psi;
    if (!is_g) return;
    double lam[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) lam[i] = 0.0;
    for (int k = N - 1; k >= 0; k--) {
#pragma unroll
        for (int i = 0; i < NX; i++) lam[i] += get(k, i);
        double gu[2], lz[NZ];
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            double acc = get(k, NX + jj);
#pragma unroll
            for (int i = 0; i < NX; i++)
                acc = fma(sens_is_const<MODEL>(NZ + jj, i) ? sens_const(NZ + jj, i) : get(k, NX + 2 + (NZ + jj) * NX + i), lam[i], acc);
            gu[jj] = acc;
        }
#pragma unroll
        for (int jj = 0; jj < NZ; jj++) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < NX; i++)
                acc = fma(sens_is_const<MODEL>(jj, i) ? sens_const(jj, i) : get(k, NX + 2 + jj * NX + i), lam[i], acc);
            lz[jj] = acc;
        }
#pragma unroll
        for (int jj = 0; jj < NZ; jj++) lam[2 + jj] = lz[jj];
        grow[2 * k] = gu[0];
        grow[2 * k + 1] = gu[1];
    }
}
template <int MODEL, class Get>
__device__ __forceinline__ void adjoint_rec(const DevCfg &c, const Workspace &w, int a, bool ch2, bool is_g, Get get)
{
    double *psi_out = w.psi_direct ? w.psi_direct + a : !ch2 ? w.rec + (size_t)a * REC + R_PSIE : nullptr;
    adjoint_rec_to<MODEL>(c, is_g, get, psi_out, (ch2 ? w.ge2 : w.ge) + (size_t)a * c.n);
}

// The same recursion for the kinematic model (NX = 4) by a QUAD of lanes per request: lane c keeps component c of the
// adjoint, lanes 0 / 1 form the two input gradients of a stage, lanes 2 / 3 the two non-position adjoint components
// (the position components only collect dL/dx); a stage is one add, four quad broadcasts and a chain of four fma per
// lane instead of ~45 dependent-issue instructions on one lane.  Every sum is formed from the same operands in the
// same order as adjoint_rec_to forms it: same bits (the fused kernel uses this form, the two-kernel path and the
// persistent kernel the one-lane form: test_wide_rollout_is_bit_identical and the MPC_UNFUSED_EVAL suite compare them).
// Why: the fused K1b+K1c workgroup holds its 44 KB of stage records in LDS for as long as this recursion runs on ONE
// wave of its four -- half of the workgroup's life (profiles/r03_experiments.txt 29) -- and three such workgroups are all
// a CU holds.  All four lanes of a quad must call it together (same request).
template <class Get>
__device__ __forceinline__ void adjoint_rec_quad_kin(const DevCfg &c, bool is_g, int comp, Get get, double *psi_out, double *grow)
{
    constexpr int NX = 4, NZ = 2, JS = JacRec<KIN>::SIZE;
    const int N = c.N;
    if (comp == 0) {
        double psi = 0.0;
        for (int k = 0; k < N; k++) psi += get(k, JS);
        if (psi_out) *psi_out = psi;
    }
    if (!is_g) return;                                   // (uniform within the quad)
    const int dd = (comp + 2) & 3;                       // the row of the sensitivity block this lane multiplies by
    double lam = 0.0;
    for (int k = N - 1; k >= 0; k--) {
        lam += get(k, comp);
        const double l0 = quad_bcast<0x00>(lam), l1 = quad_bcast<0x55>(lam), l2 = quad_bcast<0xAA>(lam), l3 = quad_bcast<0xFF>(lam);
        double acc = comp < 2 ? get(k, NX + comp) : 0.0;
        // row dd of the block; its constant entries are not stored (sens_is_const): heading row (dd = 0) [., ., 1, 0],
        // steering row (dd = 3) [., ., ., 0]
        const double t0 = get(k, NX + 2 + dd * NX + 0), t1 = get(k, NX + 2 + dd * NX + 1);
        const double t2 = dd == 0 ? 1.0 : get(k, NX + 2 + dd * NX + 2);
        const double t3 = (dd == 0 || dd == 3) ? 0.0 : get(k, NX + 2 + dd * NX + 3);
        acc = fma(t0, l0, acc);
        acc = fma(t1, l1, acc);
        acc = fma(t2, l2, acc);
        acc = fma(t3, l3, acc);
        if (comp < 2) grow[2 * k + comp] = acc;          // the input gradients of the stage
        else lam = acc;                                  // lam[2 + jj] = lz[jj]
    }
    static_assert(sens_is_const<KIN>(0, 2) && sens_is_const<KIN>(0, 3) && sens_is_const<KIN>(3, 3) && !sens_is_const<KIN>(3, 2) &&
                  !sens_is_const<KIN>(1, 3) && !sens_is_const<KIN>(2, 3) && !sens_is_const<KIN>(0, 1), "constant entries of the kinematic block");
    (void)NZ;
}

// K1b.  With `w.arrive` set it also does K1c for a block of 64 slots, in the stage-block that finishes
// last (arrival counter per slot block): the records are stored write-through (sc1: relaxed
// agent-scope atomic stores, so no release fence), the wave drains its stores, one lane adds to the
// counter, and the wave whose add returns N - 1 -- every other stage of these slots has then drained
// its stores before its own add -- invalidates its L1 (agent-scope acquire) and runs the adjoint
// recursion of its 64 slots with plain loads (the recipe of cdna_hip_programming.md, Guideline 16 R1 in
// its counter form).  No wave waits for another one.  The adjoint launch and its place in a round's
// chain of dependent launches disappear; the arithmetic is adjoint_rec either way: same bits.
// (kinematic: held to 168 registers = three waves per SIMD, 144 B of scratch per lane: K1b -3.6 %)
#ifndef MPC_K1B_WAVES
#define MPC_K1B_WAVES 3
#endif

// (tried: the gradient blocks and the cost blocks by kernels of their own -- the cost-only variant needs 55
// registers and runs eight waves per SIMD, 13 us per launch against 75 us for the gradient blocks -- but
// the pair of launches is slower than the one: 180.2 vs 175.6 ms per solve)
template <int MODEL, bool SHARED_CL>
__global__ void __launch_bounds__(64, (MODEL == KIN ? MPC_K1B_WAVES : MPC_K1B_WAVES_PAC))
stage_kernel(const DevCfg c, const Workspace w, const int *__restrict__ counts, int nG_imm, int nC_imm,
             int nblk_max)
{
    constexpr int NX = ModelDim<MODEL>::NX, JS = JacRec<MODEL>::SIZE;
#if MPC_DEV_STAMP == 6
    DevStamp stamp(blockIdx.x);
#endif
    const SlotMap sm(counts, nG_imm, nC_imm);
    const int k = blockIdx.x / nblk_max, sb = blockIdx.x % nblk_max;
    if (sb >= sm.nblk) return;
    const bool is_g = sb < sm.nblk_g;
    const int uslot = sb * 64 + threadIdx.x;
    const int raw = w.agent_of[uslot];
    const bool arr = w.arrive != nullptr;                // uniform
    // grid search on a shared centerline: the wave keeps the row's points (1.6 KB) in LDS, so that the
    // candidate points and the three geometry points after them are LDS reads, not two more trips to L2
    extern __shared__ double2 s_xy[];
    const bool lds_xy = SHARED_CL && w.near.gmeta != nullptr && c.S <= GRID_LDS_MAX_S; // uniform; the host sizes the LDS
    if (lds_xy) {
        const double2 *__restrict__ gp = reinterpret_cast<const double2 *>(w.near.gxy);
        for (int j = threadIdx.x; j < c.S; j += 64) s_xy[j] = gp[j];
        __builtin_amdgcn_wave_barrier();                                           // (all 64 lanes are still here)
    }
    if (raw < 0 && !arr) return;
    const int a = raw & AGENT_MASK;
    const bool ch2 = (raw & CH2_BIT) != 0;   // speculative channel: only the gradient is kept
    const size_t St = (size_t)w.St;
    if (raw >= 0) {
        double xs[NX], xe[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) {
            xs[i] = w.trajx[(size_t)(k * NX + i) * St + uslot];
            xe[i] = w.trajx[(size_t)((k + 1) * NX + i) * St + uslot];
        }
        const double d = w.useq[(size_t)(2 * k) * St + uslot], dl = w.useq[(size_t)(2 * k + 1) * St + uslot];
        const double *__restrict__ clp = SHARED_CL ? w.cl : w.cl + (size_t)w.cl_index[a] * 2 * (size_t)c.S;
        double *const jr = w.jac + (size_t)k * JS * St + uslot;
        double *const sl = w.stage_L + (size_t)k * St + uslot;
        const auto put = [=](int f, double v) {
            double *p = f == JS ? sl : jr + (size_t)f * St;
            if (arr) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *p = v;
        };
        if (is_g) stage_sens_record<MODEL>(c, xs, xe, d, dl, put);
        Geom g;
        if (lds_xy) {
            const int idx = nearest_index_grid(c, clp, w.near.gmeta, w.near.gcells, [=](int i) { return s_xy[i]; }, xe[0], xe[1]);
            load_geom_xy(s_xy, idx, g);
        } else {
            stage_geom(c, w, clp, SHARED_CL ? 0 : w.cl_index[a], xe[0], xe[1], g);
        }
        stage_record<MODEL>(c, w, a, ch2, is_g, k, xs, xe, d, dl, g, put);
    }
    if (!arr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's record stores have left
    int old = 0;
    if (threadIdx.x == 0) old = __hip_atomic_fetch_add(&w.arrive[sb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != c.N - 1) return;                          // uniform: not the last stage of this slot block
    if (threadIdx.x == 0) __hip_atomic_store(&w.arrive[sb], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // for the next launch
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's stale L1 lines of the record block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (raw < 0) return;
    const double *const jac = w.jac + uslot, *const slp = w.stage_L + uslot;
    adjoint_rec<MODEL>(c, w, a, ch2, is_g, [=](int kk, int f) {
        return f == JS ? slp[(size_t)kk * St] : jac[((size_t)kk * JS + f) * St];
    });
}

// (tried: four lanes per gradient request -- lane `role` of a DPP quad owning lambda[role] and one row of the
// sensitivity block, lambda exchanged by quad broadcasts, the same fma chains, bit-identical, 6 loads per
// lane and stage instead of 19 and four times the waves -- 37.8 -> 47.9 us per launch: K1c is bound by the
// record bytes coming back (104 MB per launch, written by K1b a moment before), not by its dependent trips)
template <int MODEL>
__global__ void __launch_bounds__(64)
adjoint_kernel(const DevCfg c, const Workspace w, const int *__restrict__ counts, int nG_imm, int nC_imm,
               int *__restrict__ desc)
{
    constexpr int JS = JacRec<MODEL>::SIZE;
    const SlotMap sm(counts, nG_imm, nC_imm);
    const int sb = blockIdx.x;
    // for the next step kernel's thread-per-agent blocks (chain_block): the gradient slots of this round
    if (desc && sb == 0 && threadIdx.x == 0) *desc = sm.gpad;
    if (sb >= sm.nblk) return;
    const bool is_g = sb < sm.nblk_g;
    const int uslot = sb * 64 + threadIdx.x;
    const int raw = w.agent_of[uslot];
    if (raw < 0) return;
    const size_t St = (size_t)w.St;
    const double *const jac = w.jac + uslot, *const sl = w.stage_L + uslot;
    adjoint_rec<MODEL>(c, w, raw & AGENT_MASK, (raw & CH2_BIT) != 0, is_g, [=](int k, int f) {
        return f == JS ? sl[(size_t)k * St] : jac[((size_t)k * JS + f) * St];
    });
}

// K1b + K1c in one launch: a workgroup takes SPB = BLK / N consecutive slots; thread (k, j) does stage
// k of slot j exactly as stage_kernel does, but leaves the stage record in LDS; after a barrier the
// first SPB threads run the cost sum and the adjoint recursion of their slot out of LDS.  The
// N (NX^2 + NX + 2)-double record block of a gradient request (3.5 KB at N = 20, nx = 4) never goes to
// HBM and one launch per round disappears.  Threads are stage-major (j fastest), so a wave reads
// SPB consecutive slots per stage from the slot-indexed scratch.
template <int MODEL> struct FusedBlk { static constexpr int BLK = MODEL == PAC ? 128 : 256; };

#ifndef MPC_FUSED_WAVES
#define MPC_FUSED_WAVES 3
#endif
template <int MODEL, bool SHARED_CL>
__global__ void __launch_bounds__(FusedBlk<MODEL>::BLK, (MODEL == KIN ? MPC_FUSED_WAVES : 2))
stage_adjoint_kernel(const DevCfg c, const Workspace w, const int *__restrict__ counts, int nG_imm, int nC_imm,
                     int *__restrict__ desc)
{
    constexpr int NX = ModelDim<MODEL>::NX, JS = JacRec<MODEL>::SIZE, BLK = FusedBlk<MODEL>::BLK;
    extern __shared__ double s_rec[];                    // [JS + 1][N][SPB]; row JS = stage cost
#if MPC_DEV_STAMP == 2
    DevStamp stamp(blockIdx.x * (BLK / 64) + (threadIdx.x >> 6));
#endif
    const SlotMap sm(counts, nG_imm, nC_imm);
    const int N = c.N, SPB = BLK / N;
    const int slot0 = blockIdx.x * SPB, nslots = sm.nblk * 64;
    if (desc && blockIdx.x == 0 && threadIdx.x == 0) *desc = sm.gpad;   // see adjoint_kernel
    if (slot0 >= nslots) return;
    const size_t St = (size_t)w.St;
    const int NS = N * SPB;
    {
        const int k = threadIdx.x / SPB, j = threadIdx.x - k * SPB;
        const int uslot = slot0 + j;
        const int raw = (k < N && uslot < nslots) ? w.agent_of[uslot] : -1;
        if (raw >= 0) {
            const bool is_g = uslot < sm.gpad;
            const int a = raw & AGENT_MASK;
            const bool ch2 = (raw & CH2_BIT) != 0;   // speculative channel: only the gradient is kept
            double xs[NX], xe[NX];
#pragma unroll
            for (int i = 0; i < NX; i++) {
                xs[i] = w.trajx[(size_t)(k * NX + i) * St + uslot];
                xe[i] = w.trajx[(size_t)((k + 1) * NX + i) * St + uslot];
            }
            const double d = w.useq[(size_t)(2 * k) * St + uslot], dl = w.useq[(size_t)(2 * k + 1) * St + uslot];
            const double *__restrict__ clp = SHARED_CL ? w.cl : w.cl + (size_t)w.cl_index[a] * 2 * (size_t)c.S;
            double *const r = s_rec + k * SPB + j;
            const auto put = [=](int f, double v) { r[(size_t)f * NS] = v; };
            if (is_g) stage_sens_record<MODEL>(c, xs, xe, d, dl, put);
            Geom g;
            stage_geom(c, w, clp, SHARED_CL ? 0 : w.cl_index[a], xe[0], xe[1], g);
            stage_record<MODEL>(c, w, a, ch2, is_g, k, xs, xe, d, dl, g, put);
        }
    }
    __syncthreads();
    if constexpr (MODEL == KIN) {
        // the adjoint recursion by a quad of lanes per request (adjoint_rec_quad_kin): 4 SPB lanes -- more than the
        // workgroup has threads when the horizon is shorter than four stages, hence the loop (quads stay aligned: 4 | BLK)
        for (int t = threadIdx.x; t < 4 * SPB; t += BLK) {
            const int j = t >> 2, comp = t & 3, uslot = slot0 + j;
            const int raw = uslot < nslots ? w.agent_of[uslot] : -1;
            if (raw < 0) continue;                       // (the same for the four lanes of a quad)
            const double *const rj = s_rec + j;
            const int a = raw & AGENT_MASK;
            const bool ch2 = (raw & CH2_BIT) != 0;
            double *psi_out = w.psi_direct ? w.psi_direct + a : !ch2 ? w.rec + (size_t)a * REC + R_PSIE : nullptr;
            adjoint_rec_quad_kin(c, uslot < sm.gpad, comp, [=](int k, int f) { return rj[(size_t)f * NS + k * SPB]; },
                                 psi_out, (ch2 ? w.ge2 : w.ge) + (size_t)a * c.n);
        }
        return;
    }
    if ((int)threadIdx.x >= SPB) return;
    const int j = threadIdx.x, uslot = slot0 + j;
    const int raw = uslot < nslots ? w.agent_of[uslot] : -1;
    if (raw < 0) return;
    const double *const rj = s_rec + j;
    adjoint_rec<MODEL>(c, w, raw & AGENT_MASK, (raw & CH2_BIT) != 0, uslot < sm.gpad,
                       [=](int k, int f) { return rj[(size_t)f * NS + k * SPB]; });
}

} // namespace mpc
