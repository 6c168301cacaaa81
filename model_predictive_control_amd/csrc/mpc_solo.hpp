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
// Lookahead (Pacejka model, N <= 16, no constraints): FOUR evaluations side by side, 16 lanes each -- the one the state
// machine asked for, its speculative gradient, and candidates the state machine is going to ask for if the test it is
// about to make fails (the next line-search trial points; the prox points of the next descent-lemma doublings),
// kept in a small cache of (point -> psi, gradient) and handed back WITHOUT an evaluation trip when asked for.
// A Pacejka evaluation uses 4 lanes for the serial rollout and N for the stage records: the other lanes of the lone
// wave are idle, and the tail of a batch is one agent's chain of dependent trips (VERDICT r3 item 4,
// profiles/r03_pacejka_tail_model.txt).  psi and grad psi are functions of the point alone (m = 0), so a cached
// evaluation is THE evaluation: results, iteration and evaluation counts are those of the plain kernel
// (MPC_NO_LOOKAHEAD; tests), only trips are saved.
constexpr int LA_SLOTS = 4, LA_LANES = 16, LA_ENTRIES = 8;
__host__ __device__ inline bool solo_lookahead(int model, int nfe, int N, int m, int no_la)
{
    return model == PAC && !no_la && m == 0 && N <= LA_LANES && solo_dual(model, nfe, N);
}
struct LaSlot { const double *in; double *grad; double *psi; int live, is_g; };   // what the 16 lanes of a slot evaluate
// doubles of LDS the cache and the slot descriptors take
__host__ __device__ inline size_t la_lds_doubles(int n) { return (size_t)LA_ENTRIES * (2 * n + 2) + LA_SLOTS * 4 + 8; }

template <int MODEL> __host__ __device__ inline size_t solo_lds_doubles(int nfe, int N, int n, int M, bool hist, bool la = false)
{
    constexpr int NX = ModelDim<MODEL>::NX, JS = JacRec<MODEL>::SIZE;
    const size_t per = (size_t)(N + 1) * NX + (size_t)(JS + 1) * N;
    return (hist ? (size_t)2 * M * n : 0) + (la ? LA_SLOTS : solo_dual(MODEL, nfe, N) ? 2 : 1) * per + (la ? la_lds_doubles(n) : 0);
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
    if constexpr (MODEL == KIN) {
        // (the kinematic adjoint by a quad of lanes: mpc_eval.hpp adjoint_rec_quad_kin -- same bits, a third of the chain)
        if (hl < 4 && live) {
            double *psi_out = w.psi_direct ? w.psi_direct + a : !ch2 ? w.rec + (size_t)a * REC + R_PSIE : nullptr;
            adjoint_rec_quad_kin(c, is_g, hl, [=](int k, int f) { return rc[f * N + k]; }, psi_out,
                                 (ch2 ? w.ge2 : w.ge) + (size_t)a * n);
        }
    } else if (hl == 0 && live) adjoint_rec<MODEL>(c, w, a, ch2, is_g, [=](int k, int f) { return rc[f * N + k]; });
    SOLO_CLK(adj, tclk);
}

// ---- lookahead: the evaluation of up to four points by one wave (Pacejka), slot s = lanes 16 s .. 16 s + 15
__device__ __forceinline__ void solo_eval_la(const DevCfg &c, const Workspace &w, int a, int lane, const LaSlot *slots,
                                             double *traj, double *rec)
{
    constexpr int NX = ModelDim<PAC>::NX, JS = JacRec<PAC>::SIZE;
    const int N = c.N;
    const int sl = lane >> 4, hl = lane & 15;
    const LaSlot d = slots[sl];
    const bool live = d.live != 0, is_g = d.is_g != 0;
    const double *__restrict__ row = d.in;
    const bool stage_lane = live && hl < N;
    const double dv = stage_lane ? row[2 * hl] : 0.0, dl = stage_lane ? row[2 * hl + 1] : 0.0;
    double *const tj = traj + (size_t)sl * ((size_t)(N + 1) * NX);
    double *const rc = rec + (size_t)sl * ((size_t)(JS + 1) * N);
    double x0[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) x0[i] = w.x0[(size_t)a * NX + i];
    if (hl < 4 && live) {                                  // the serial recurrence on a quad of lanes (rhs_quad)
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
    __builtin_amdgcn_wave_barrier();                       // LDS is in order within a wave
    if (stage_lane) {
        double xs[NX], xe[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) { xs[i] = tj[hl * NX + i]; xe[i] = tj[(hl + 1) * NX + i]; }
        const double *__restrict__ clp = w.cl_index ? w.cl + (size_t)w.cl_index[a] * 2 * (size_t)c.S : w.cl;
        double *const r = rc + hl;
        const auto put = [=](int f, double v) { r[f * N] = v; };
        if (is_g) stage_sens_record<PAC>(c, xs, xe, dv, dl, put);
        Geom g;
        stage_geom(c, w, clp, w.cl_index ? w.cl_index[a] : 0, xe[0], xe[1], g);
        stage_record<PAC>(c, w, a, /*ch2 (unused: m = 0)=*/true, is_g, hl, xs, xe, dv, dl, g, put);
    }
    __builtin_amdgcn_wave_barrier();
    if (hl == 0 && live) adjoint_rec_to<PAC>(c, is_g, [=](int k, int f) { return rc[f * N + k]; }, d.psi, d.grad);
    __builtin_amdgcn_wave_barrier();
}

// the cache: LA_ENTRIES x (point[n], gradient[n]) + psi[LA_ENTRIES] + flags[LA_ENTRIES] (0 empty, 1 psi, 2 psi and gradient)
struct LaCache {
    double *pt, *gr, *psi; int *fl; int n;
    __device__ __forceinline__ LaCache(double *base, int n_) : pt(base), gr(base + (size_t)LA_ENTRIES * n_),
        psi(base + (size_t)2 * LA_ENTRIES * n_), fl((int *)(base + (size_t)2 * LA_ENTRIES * n_ + LA_ENTRIES)), n(n_) {}
    __device__ __forceinline__ void clear(int lane) { if (lane < LA_ENTRIES) fl[lane] = 0; }
    // the entry that holds the evaluation of the point whose element `lane` is v (lanes >= n: ignored), or -1; uniform
    __device__ __forceinline__ int find(int lane, double v, bool need_grad) const
    {
        for (int e = 0; e < LA_ENTRIES; e++) {
            const int f = __builtin_amdgcn_readfirstlane(fl[e]);
            if (f == 0 || (need_grad && f != 2)) continue;
            const bool eq = lane >= n || __double_as_longlong(pt[(size_t)e * n + lane]) == __double_as_longlong(v);
            if (__ballot(!eq) == 0ull) return e;
        }
        return -1;
    }
};

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

// One agent's solve with the lookahead (Pacejka, NE = 1): the loop of solo_kernel with a cache lookup before every
// evaluation trip and candidates riding in the free slots of the trip.
template <int MC>
__device__ __forceinline__ void solo_agent_la(const DevCfg &c, const Workspace &w, int a, int lane, double *hist,
                                              double *traj, double *rec, double *la_base, long long max_trips)
{
#pragma clang fp contract(off)   // the candidate points are formed by the state machine's own functions: same roundings
    const int n = c.n, par = lane & 1;
    const size_t an = (size_t)a * n;
    LaCache cache(la_base, n);
    LaSlot *slots = (LaSlot *)(la_base + (size_t)LA_ENTRIES * (2 * n + 2));
    static_assert(sizeof(LaSlot) == 4 * sizeof(double), "slot descriptor layout");
    cache.clear(lane);
    int next_e = 0;                                       // round-robin replacement (uniform)
    double la_evals = 0.0, la_hits = 0.0;
    __builtin_amdgcn_wave_barrier();
    const auto ldv = [&](const double *rowp) { return lane < n ? rowp[lane] : 0.0; };
    for (long long trip = 0; trip < max_trips; trip++) {
        const AgentIn<1> in = load_agent<1>(c, w, a, lane);
        int req = advance_agent<1, MC>(c, w, a, lane, in, hist, false, /*allow_spec=*/true);
        if ((req & (REQ_GRAD | REQ_COST)) == 0) break;                // uniform: the agent is done
        // the record as the state machine left it (same wave, same addresses: ordered behind its stores)
        double rv = w.rec[(size_t)a * REC + lane];
        const int ph = __builtin_amdgcn_readlane(__double2loint(rv), R_PHASE) & PH_MASK;
        const double gam = rdlane(rv, R_GAMMA), gamn = rdlane(rv, R_GAMMAN), tau = rdlane(rv, R_TAU);
        const double Lk = rdlane(rv, R_L), Ln = rdlane(rv, R_LN);
        const bool want_g = (req & REQ_GRAD) != 0;
        bool need_main = true, need_spec = (req & REQ_SPEC) != 0;
        // ---- a descent-lemma loop that has doubled twice already: the speculative gradient of every further level
        // would be thrown away but the last one's; it is dropped (the next iteration asks for its Hessian-vector
        // gradient itself: same bits, same counted evaluations) and the slots go to deeper levels
        const bool dl_loop = (ph == PH_W_LS_C && Ln >= 4.0 * Lk) || (ph == PH_W_DL && rdlane(rv, R_GAMMA) < rdlane(rv, R_GAMMA_TOP) * 0.3);
        if (need_spec && dl_loop) {
            need_spec = false; req &= ~REQ_SPEC;
            if (lane == R_SPEC) w.rec[(size_t)a * REC + lane] = rec_int(0);
            if (lane == R_NSPEC || lane == R_NGRAD) w.rec[(size_t)a * REC + lane] = rv - 1.0;
        }
        // ---- served from the cache?
        const double xe_v = ldv(w.xe + an);
        {
            const int e = cache.find(lane, xe_v, want_g);
            if (e >= 0) {
                if (lane == 0) w.rec[(size_t)a * REC + R_PSIE] = cache.psi[e];
                if (want_g && lane < n) w.ge[an + lane] = cache.gr[(size_t)e * n + lane];
                need_main = false;
            }
        }
        if (need_spec) {
            const int e = cache.find(lane, ldv(w.xe2 + an), true);
            if (e >= 0) {
                if (lane < n) w.ge2[an + lane] = cache.gr[(size_t)e * n + lane];
                need_spec = false;
            }
        }
        if (!need_main && !need_spec) { la_hits += 1.0; continue; }   // no trip
        // ---- the slots of this trip
        int nslot = 0;
        const auto set_slot = [&](int s, const double *inp, double *g, double *ps, bool isg) {
            if (lane == 0) { slots[s].in = inp; slots[s].grad = g; slots[s].psi = ps; slots[s].live = 1; slots[s].is_g = isg ? 1 : 0; }
        };
        if (lane < LA_SLOTS) slots[lane].live = 0;
        __builtin_amdgcn_wave_barrier();
        if (need_main) set_slot(nslot++, w.xe + an, w.ge + an, w.rec + (size_t)a * REC + R_PSIE, want_g);
        if (need_spec) set_slot(nslot++, w.xe2 + an, w.ge2 + an, nullptr, true);
        // a candidate: the point (element `lane` = v) goes into a cache entry unless the cache holds it already
        int cand_e[LA_SLOTS], cand_g[LA_SLOTS], ncand = 0;
        const auto add_candidate = [&](double v, bool isg) {
            if (nslot >= LA_SLOTS) return;
            if (__ballot(lane < n && !isfinite(v)) != 0ull) return;       // (uniform) nothing to learn from a lost point
            if (cache.find(lane, v, isg) >= 0) return;
            if (__ballot(lane < n && __double_as_longlong(v) != __double_as_longlong(xe_v)) == 0ull && need_main && (want_g || !isg)) return;
            const int e = next_e; next_e = (next_e + 1) % LA_ENTRIES;
            if (lane == 0) cache.fl[e] = 0;
            if (lane < n) cache.pt[(size_t)e * n + lane] = v;
            set_slot(nslot++, cache.pt + (size_t)e * n, cache.gr + (size_t)e * n, cache.psi + e, isg);
            cand_e[ncand] = e; cand_g[ncand] = isg ? 1 : 0; ncand++;
        };
        Row<1> X, G, Q, XN, GE;
        if (ph == PH_W_LS_G) {
            // waiting for the gradient at the trial point of step tau: the next trials are tau / 2, tau / 4, tau / 8
            // while they are >= tau_min (PH_W_LS_C), the last of them the safe prox step (PH_LS_TRIAL)
            X.v[0] = ldv(w.xk + an); G.v[0] = ldv(w.gk + an); Q.v[0] = ldv(w.q + an);
            double t = tau;
            for (int j = 0; j < 3; j++) {
                t = t / 2.0;
                if (!(t >= c.tau_min)) break;
                const Row<1> xt = trial_point<1>(c, par, X, G, Q, gam, t, t / 2.0 < c.tau_min);
                add_candidate(xt.v[0], true);
            }
        } else if (ph == PH_W_LS_C || ph == PH_W_DL) {
            // waiting for the cost at a prox point xhat(gamma_n) of (x, g) = the trial point and its gradient (W_LS_C) or
            // the iterate (W_DL).  If the descent lemma fails there the step is halved: xhat(gamma_n / 2), / 4, ...
            const bool at_trial = ph == PH_W_LS_C;
            XN.v[0] = ldv((at_trial ? w.xn : w.xk) + an); GE.v[0] = ldv((at_trial ? w.ge : w.gk) + an);
            const double g0 = at_trial ? gamn : gam;
            const double Lcur = at_trial ? Ln : Lk;
            if (!dl_loop && at_trial) {
                // not in a doubling loop: the next trial's cost and speculative gradient, if its gradient is known
                const double t = tau / 2.0;
                if (t >= c.tau_min) {
                    X.v[0] = ldv(w.xk + an); G.v[0] = ldv(w.gk + an); Q.v[0] = ldv(w.q + an);
                    const Row<1> xt = trial_point<1>(c, par, X, G, Q, gam, t, t / 2.0 < c.tau_min);
                    const int e = cache.find(lane, xt.v[0], true);
                    if (e >= 0) {
                        Row<1> gt; gt.v[0] = lane < n ? cache.gr[(size_t)e * n + lane] : 0.0;
                        const double xh = xt.v[0] + prox_p(c, par, xt.v[0], gt.v[0], gam);       // prox_to_xe at the next trial
                        add_candidate(xh, false);
                        Row<1> sp; sp.v[0] = 0.0;
                        const int nj = spec_point<1>(c, par, n, lane, xt, gt, gam, sp);
                        if (!c.no_spec && nj > 0 && nj < n) add_candidate(sp.v[0], true);
                    }
                }
            }
            // deeper levels of this point's descent-lemma loop fill what is left
            double gd = g0, Ld = Lcur;
            for (int j = 0; j < LA_SLOTS; j++) {
                if (nslot >= LA_SLOTS || !(Ld * 2.0 <= c.L_max)) break;
                gd = gd / 2.0; Ld = Ld * 2.0;
                add_candidate(XN.v[0] + prox_p(c, par, XN.v[0], GE.v[0], gd), false);
            }
        }
        __builtin_amdgcn_wave_barrier();
        solo_eval_la(c, w, a, lane, slots, traj, rec);
        for (int j = 0; j < ncand; j++) if (lane == 0) cache.fl[cand_e[j]] = cand_g[j] ? 2 : 1;
        la_evals += (double)ncand;
        __builtin_amdgcn_wave_barrier();
    }
    // the lookahead's statistics of this agent (plain double counters of the record)
    if (lane == R_LA_EVALS) w.rec[(size_t)a * REC + lane] += la_evals;
    if (lane == R_LA_HITS) w.rec[(size_t)a * REC + lane] += la_hits;
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

// LA: the lookahead variant (host: solo_lookahead) -- a kernel of its own, so that neither holds the other's code
template <int MODEL, int NE, int MC, bool LA = false>
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
    if constexpr (LA) {
        // (one element per lane, one wave per workgroup; four trajectories, then four record blocks, then the cache)
        static_assert(!LA || (MODEL == PAC && NE == 1 && SOLO_WAVES == 1), "lookahead: Pacejka, n <= 64, one wave per workgroup");
        constexpr int NXm = ModelDim<MODEL>::NX, JSm = JacRec<MODEL>::SIZE;
        rec = traj + (size_t)LA_SLOTS * ((size_t)(c.N + 1) * NXm);
        double *la_base = rec + (size_t)LA_SLOTS * ((size_t)(JSm + 1) * c.N);
        for (;;) {
            int i = 0;
            if (lane == 0) i = atomicAdd(&ctr[0], 1);
            i = __builtin_amdgcn_readfirstlane(i);
            if (i >= total) break;
            solo_agent_la<MC>(c, w, list ? list[i] : i, lane, hist, traj, rec, la_base, max_trips);
        }
        return;
    }
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
