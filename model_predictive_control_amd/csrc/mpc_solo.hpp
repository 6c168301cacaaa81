// mpc_solo.hpp -- the whole solve of an agent inside ONE wavefront, without returning to the host.
//
// A round of the batched path (step_kernel -> K1a -> K1b -> K1c) costs four dependent launches
// whatever the number of agents in it; once few agents are left that is pure latency (a tail round
// with a handful of requests still takes ~55 us), and the slowest agent of a batch needs hundreds of
// rounds more than the average one.  The persistent kernel below takes over when a sub-batch group has
// few running agents left (and serves small batches from the start): a wave claims a running agent,
// then loops  advance_agent -> evaluation -> advance_agent ...  until the agent is done, and claims
// the next one.  Nothing is exchanged between waves, so no grid-wide synchronisation is needed and
// every wave reaches its exit (the claim counter runs out; an agent's state machine is bounded by the
// solver's own iteration budgets, and a trip limit guards the loop besides).
//
// The evaluation inside the wave reuses the device functions of the round path with the same
// roundings -- the wave-per-request rollout (kin_wide_rollout) for the kinematic model, the
// thread-per-agent stage_forward on one lane otherwise; stage k of the horizon on lane k
// (stage_record); the adjoint recursion on one lane (adjoint_rec) -- so an agent gets the
// same bits whichever path serves it, and the host may switch on the live request counts.
// Speculative gradients: with the lane-serial rollout (Pacejka model) the speculative request of a
// round is evaluated beside the regular one, on the other half of the wave (solo_dual); with the
// wave-per-request rollout (kinematic model) two evaluations would run one after the other, so none is
// issued -- one that is pending when the agent arrives is consumed either way.
#pragma once
#include "mpc_eval.hpp"

namespace mpc {

constexpr int SOLO_WAVES = 1; // one wave per workgroup: nothing is shared between waves, and the dispatcher places them freely

// Two evaluations of one agent side by side in its wave (the request of the round on lanes 0 .. 31,
// the speculative gradient on lanes 32 .. 63): possible when the rollout is the lane-serial one (the
// wave-per-request kinematic rollout needs all 64 lanes) and a horizon fits half a wave.  With it the
// persistent kernel keeps the round path's speculation for free: both serial chains advance in the same
// instructions.
__host__ __device__ inline bool solo_wide(int model, int nfe) { return model == KIN && nfe == 4; }
__host__ __device__ inline bool solo_dual(int model, int nfe, int N) { return !solo_wide(model, nfe) && N <= 32; }

// doubles of LDS one wave needs: history copy (MC < 0), trajectory, stage records (twice when dual)
template <int MODEL> __host__ __device__ inline size_t solo_lds_doubles(int nfe, int N, int n, int M, bool hist)
{
    constexpr int NX = ModelDim<MODEL>::NX, JS = JacRec<MODEL>::SIZE;
    const size_t per = (size_t)(N + 1) * NX + (size_t)(JS + 1) * N;
    return (hist ? (size_t)2 * M * n : 0) + (solo_dual(MODEL, nfe, N) ? 2 : 1) * per;
}

// the evaluation(s) agent `a` asked for (req: REQ_GRAD or REQ_COST, plus REQ_SPEC) by its whole wave
#if MPC_DEV_STAMP == 5
struct SoloClk { long long roll = 0, recs = 0, adj = 0; };
#define SOLO_CLK_ARG , SoloClk &clk
#define SOLO_CLK(field, t) do { const long long now_ = __builtin_amdgcn_s_memrealtime(); clk.field += now_ - t; t = now_; } while (0)
#else
#define SOLO_CLK_ARG
#define SOLO_CLK(field, t) do { } while (0)
#endif
template <int MODEL>
__device__ __forceinline__ void solo_eval(const DevCfg &c, const Workspace &w, int a, int lane, int req,
                                          double *traj, double *rec SOLO_CLK_ARG)
{
#if MPC_DEV_STAMP == 5
    long long tclk = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int NX = ModelDim<MODEL>::NX, JS = JacRec<MODEL>::SIZE;
    const int N = c.N, n = c.n;
    const bool wide = solo_wide(MODEL, c.nfe), dual = solo_dual(MODEL, c.nfe, N);   // uniform
    const int half = dual ? lane >> 5 : 0, hl = dual ? lane & 31 : lane;
    const bool ch2 = half == 1;                                     // the speculative channel: xe2 -> ge2 only
    const bool live = !ch2 || (req & REQ_SPEC) != 0;
    const bool is_g = ch2 || (req & REQ_GRAD) != 0;
    const double *__restrict__ row = (ch2 ? w.xe2 : w.xe) + (size_t)a * n;
    const bool stage_lane = live && hl < N;
    const double d = stage_lane ? row[2 * hl] : 0.0, dl = stage_lane ? row[2 * hl + 1] : 0.0;
    double *const tj = traj + (size_t)half * ((size_t)(N + 1) * NX);
    double *const rc = rec + (size_t)half * ((size_t)(JS + 1) * N);
    double x0[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) x0[i] = w.x0[(size_t)a * NX + i];
    if constexpr (MODEL == KIN) {
        if (wide) kin_wide_rollout(c, x0, d, dl, lane, [=](int k, int i, double v) { tj[k * 4 + i] = v; });
    }
    if constexpr (MODEL == PAC) {
        if (hl < 4 && live) {                              // the serial recurrence on a quad of lanes (rhs_quad)
            double x[NX];
#pragma unroll
            for (int i = 0; i < NX; i++) { x[i] = x0[i]; if (hl == 0) tj[i] = x0[i]; }
            for (int k = 0; k < N; k++) {
                StageInput<PAC> u;
                prep_input(c, row[2 * k], row[2 * k + 1], u);
                stage_forward_quad(c, u, x, hl);
                if (hl == 0) {
#pragma unroll
                    for (int i = 0; i < NX; i++) tj[(k + 1) * NX + i] = x[i];
                }
            }
        }
    } else if (!wide && hl == 0 && live) {                 // the serial recurrence, as rollout_kernel runs it
        double x[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) { x[i] = x0[i]; tj[i] = x0[i]; }
        for (int k = 0; k < N; k++) {
            StageInput<MODEL> u;
            prep_input(c, row[2 * k], row[2 * k + 1], u);
            stage_forward<MODEL>(c, u, x);
#pragma unroll
            for (int i = 0; i < NX; i++) tj[(k + 1) * NX + i] = x[i];
        }
    }
    __builtin_amdgcn_wave_barrier();                       // LDS is in order within a wave
    SOLO_CLK(roll, tclk);
    if (stage_lane) {
        double xs[NX], xe[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) { xs[i] = tj[hl * NX + i]; xe[i] = tj[(hl + 1) * NX + i]; }
        const double *__restrict__ clp = w.cl_index ? w.cl + (size_t)w.cl_index[a] * 2 * (size_t)c.S : w.cl;
        double *const r = rc + hl;
        const auto put = [=](int f, double v) { r[f * N] = v; };
        if (is_g) stage_sens_record<MODEL>(c, xs, xe, d, dl, put);
        Geom g;
        stage_geom(c, w, clp, w.cl_index ? w.cl_index[a] : 0, xe[0], xe[1], g);
        stage_record<MODEL>(c, w, a, ch2, is_g, hl, xs, xe, d, dl, g, put);
    }
    __builtin_amdgcn_wave_barrier();
    SOLO_CLK(recs, tclk);
    if (hl == 0 && live) adjoint_rec<MODEL>(c, w, a, ch2, is_g, [=](int k, int f) { return rc[f * N + k]; });
    SOLO_CLK(adj, tclk);
}

// list of the agents of this view that are still running (phase != PH_DONE), in agent order inside a
// workgroup; also resets the claim counter's companion (the number of entries)
__global__ void __launch_bounds__(256) solo_list_kernel(const Workspace w, int *__restrict__ list, int *__restrict__ ctr)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = a < w.B && rec_int_of(w.rec[(size_t)a * REC + R_PHASE]) != PH_DONE;
    const unsigned long long bal = __ballot(on);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && bal != 0ull) base = atomicAdd(&ctr[1], __popcll(bal));
    base = __builtin_amdgcn_readfirstlane(base);
    if (on) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = a;
}

// K1 alone through the wave-per-agent evaluation (standalone entry point, parity tests): agent = block
template <int MODEL>
__global__ void __launch_bounds__(64, 1) solo_eval_kernel(const DevCfg c, const Workspace w, int want_grad)
{
    extern __shared__ double s_solo[];
    double *traj = s_solo;
    double *rec = traj + (size_t)(c.N + 1) * ModelDim<MODEL>::NX;
#if MPC_DEV_STAMP == 5
    SoloClk clk;
    solo_eval<MODEL>(c, w, blockIdx.x, threadIdx.x, want_grad ? REQ_GRAD : REQ_COST, traj, rec, clk);
#else
    solo_eval<MODEL>(c, w, blockIdx.x, threadIdx.x, want_grad ? REQ_GRAD : REQ_COST, traj, rec);
#endif
}

// ctr[0] = claim counter, ctr[1] = number of list entries (list == nullptr: every agent of the view)
// Waves per SIMD the kernel is compiled for.  The kinematic variant needs 346 registers; held to 256 (two
// waves per SIMD) it spills ~90 of them and a lone wave is no slower for it (1 024 agents: 32.5 vs 33.1 ms),
// while twice as many agents are in flight (4 096 agents: 78 -> 53 ms).  The Pacejka variant needs all 512
// and its long serial chains lose more to the spills than they gain (65 536 agents: 0.97 -> 1.01 s): one wave.
// (the variant that caches twenty history pairs in registers cannot be held to 256 either)
#ifndef MPC_SOLO_WPS_KIN
#define MPC_SOLO_WPS_KIN 2
#endif
template <int MODEL, int MC> struct SoloOcc { static constexpr int WPS = (MODEL == KIN && MC <= 0) ? MPC_SOLO_WPS_KIN : 1; };

template <int MODEL, int NE, int MC>
__global__ void __launch_bounds__(64 * SOLO_WAVES, (SoloOcc<MODEL, MC>::WPS))
solo_kernel(const DevCfg c, const Workspace w, const int *__restrict__ list, int *__restrict__ ctr,
            long long max_trips)
{
    extern __shared__ double s_solo[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t per_wave = solo_lds_doubles<MODEL>(c.nfe, c.N, c.n, c.M, MC < 0);
    const bool spec = solo_dual(MODEL, c.nfe, c.N);      // speculative gradients only where they ride along
    double *hist = s_solo + (size_t)wv * per_wave;
    double *traj = hist + (MC < 0 ? (size_t)2 * c.M * c.n : 0);
    double *rec = traj + (size_t)(c.N + 1) * ModelDim<MODEL>::NX;
    const int total = list ? ctr[1] : w.B;
    for (;;) {
        int i = 0;
        if (lane == 0) i = atomicAdd(&ctr[0], 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= total) break;
        const int a = list ? list[i] : i;
#if MPC_DEV_STAMP == 5
        const long long st0 = __builtin_amdgcn_s_memrealtime();
        long long ntrip = 0, t_adv = 0;
        SoloClk clk;
#endif
        for (long long trip = 0; trip < max_trips; trip++) {
#if MPC_DEV_STAMP == 5
            const long long ta = __builtin_amdgcn_s_memrealtime();
#endif
            const AgentIn<NE> in = load_agent<NE>(c, w, a, lane);
            const int req = advance_agent<NE, MC>(c, w, a, lane, in, hist, false, /*allow_spec=*/spec);
            if ((req & (REQ_GRAD | REQ_COST)) == 0) break;              // uniform: the agent is done
#if MPC_DEV_STAMP == 5
            t_adv += __builtin_amdgcn_s_memrealtime() - ta;
            solo_eval<MODEL>(c, w, a, lane, req, traj, rec, clk);
            ntrip++;
#else
            solo_eval<MODEL>(c, w, a, lane, req, traj, rec);
#endif
        }
#if MPC_DEV_STAMP == 5
        if (lane == 0 && i < DEV_STAMPS) {   // (one buffer for all groups: the claim index of the group whose kernel ran last wins)
            g_dev_stamps[4 * i] = st0; g_dev_stamps[4 * i + 1] = __builtin_amdgcn_s_memrealtime();
            g_dev_stamps[4 * i + 2] = ntrip | (t_adv << 20); g_dev_stamps[4 * i + 3] = a | (clk.roll << 20);
            if (i + 32768 < DEV_STAMPS) { g_dev_stamps[4 * (i + 32768)] = clk.recs; g_dev_stamps[4 * (i + 32768) + 1] = clk.adj; }
        }
#endif
    }
}

} // namespace mpc
