// mpc_api.hip -- C-ABI of libmpc_hip.so (see include/mpc_hip.h): handle, workspace, launch
// orchestration of the batched MPC solve on one MI355X.  One process / one handle per GPU.
#include "../../include/mpc_hip.h"
#include "mpc_aux.hpp"
#include "mpc_solo.hpp"
#include "mpc_game.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <array>
#include <vector>

using namespace mpc;

#define MPC_MAX_GROUPS 8
#define MPC_GRID_MAX_ROWS 1024   // centerline rows the nearest-point grid is built for (256 KB each)

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(MPC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

struct mpc_handle {
    mpc_config cfg;
    DevCfg dc;
    int device = 0;
    int wide_max = 4096;        // requests per round up to which K1a runs one wave per request (MPC_WIDE_MAX)
    int apb_env = 0;            // MPC_APB: agents per step-kernel workgroup (4, 16, 64; 0 = by batch size)
    bool fused_eval = true;     // K1b + K1c in one launch (MPC_UNFUSED_EVAL: the two-kernel path)
    int fused_max = 1 << 30;    // ... while a round holds at most this many requests (MPC_FUSED_MAX).  Every round since the
                                // step kernel's workgroups hold 38 KB of LDS instead of 51 (round 3): the fused kernel's
                                // 44 KB workgroups now share a CU with them, and the stage records never leave LDS
                                // (rounds 1 - 2: 16384 -- beyond that the two-kernel path was faster)
    bool arrive_adjoint = false; // MPC_ARRIVE: K1c inside K1b's last-arriving stage block instead of a launch of its own
                                 // (same bits; measured 3 % slower -- write-through record stores: DESIGN.md 6)
    int *arrive_buf = nullptr;  // arrival counters, one per block of 64 slots
    bool quad_rollout = true;   // K1a by two (kinematic) / four (Pacejka) lanes per request (MPC_NO_QUAD: one thread)
    int pac_quad_max = 24576;   // Pacejka: requests bound of a round up to which K1a runs four lanes per request (MPC_PAC_QUAD_MAX)
    bool step_regs = false;     // MPC_STEP_REGS at mpc_create: history rows cached in registers, not LDS
    int chain_min = 24576;      // MPC_CHAIN_MIN: requests bound of a group's round from which the thread-per-agent blocks
                                // (chain_block) ride in its step launch.  Measured with the one-wave form (r03_experiments 18):
                                // 65 536 agents (groups of 16 384) +1.2 % with them; 32 768 agents (groups of 10 923)
                                // and 16 384 (groups of 8 192) -1 ... -2 %: only the full rounds of big groups
    int lds_pairs = 0;          // MPC_LDS_PAIRS: history pairs the step kernel's LDS copy holds (0 = chosen by launch_step_t)
    int num_cus = 256;
    // block bounding boxes of the centerline table last handed to mpc_centerline_blocks
    double *cl_boxes = nullptr;
    size_t cl_boxes_bytes = 0;
    const double *cl_boxes_for = nullptr; // the table they describe (device pointer identity)
    int cl_boxes_rows = 0;
    int nearest_mode = 2;                 // mpc_set_nearest_blocks: 0 full scan (MPC_NEAREST_SCAN), 1 block boxes (MPC_NEAREST_BLOCKS;
                                          // measured slower: profiles/r02_nearest_blocks.txt), 2 grid of index ranges (default);
                                          // a table that mpc_centerline_blocks has not prepared takes the full scan
    double *cl_gmeta = nullptr, *cl_gxy = nullptr;   // grid placement [C][GRID_META], interleaved points [C][S][2]
    unsigned *cl_gcells = nullptr;                   // [C][GRID_CELLS]
    int cl_grid_cap = 0;                             // rows the grid buffers hold
    bool cl_grid_ok = false;                         // the grid describes the prepared table (not built for > MPC_GRID_MAX_ROWS rows)
    int solo_all = 4096;        // a batch of at most this many agents runs in the persistent kernel from the start
                                // (MPC_SOLO_ALL; measured: kinematic 4 096 agents 62.8 -> 53.3 ms, 8 192 worse; Pacejka 1 024)
    int solo_max = 1024;        // a group with at most this many requests per round finishes in the persistent
                                // wave-per-agent kernel (MPC_SOLO_MAX / mpc_set_solo_max; 0 = rounds only).
                                // Default 1024 (kinematic model, measured in round 2, also for N = 40: profiles/r02c_*),
                                // 128 on the Pacejka model (round 4: mpc_create)
    int Bp_alloc = 0;      // workspace capacity (agents)
    char *arena = nullptr; // one device allocation carved into the Workspace arrays
    size_t arena_bytes = 0;
    Workspace ws{};
    int *host_counts = nullptr; // pinned, 512 B: [2 poll windows][MPC_MAX_GROUPS][2] ints, then (byte 128) the sixteen totals of a
                                // solve (16 x 8 B) and (byte 256) the persistent kernel's counters -- copies into pageable memory would
                                // block the host until the stream has drained, whatever the wall-clock bound says
    hipEvent_t pollev[2][MPC_MAX_GROUPS] = {{nullptr}};
    hipEvent_t soloev[MPC_MAX_GROUPS][2] = {{nullptr}}; // profile mode: around a group's persistent-kernel launch
    // profiling of the last solve
    bool profile = false;
    int64_t rounds = 0, evals_grad = 0, evals_cost = 0, launches = 0;
    double eval_ms = 0.0, step_ms = 0.0, lbfgs_ms = 0.0;
    double kernel_ms[5] = {0, 0, 0, 0, 0}; // step, K1a rollout, K1b stage, K1c adjoint, solo (profile mode)
    int64_t kernel_launches[5] = {0, 0, 0, 0, 0};
    int64_t solo_agents = 0;    // agents finished by the persistent kernel in the last solve
    double solo_longest_ms = 0.0; // profile mode: the longest of the groups' persistent-kernel launches
    int64_t spec_issued = 0, spec_used = 0; // speculative channel-2 gradients of the last solve
    int64_t la_evals = 0, la_hits = 0;      // persistent kernel's lookahead: candidate evaluations executed, requests served from them
    int64_t lbfgs_rows = 0; // history pairs read by K3 (each is read twice: 4*n*8 bytes per pair)
    std::vector<hipEvent_t> ev_pool;
    // sub-batch pipelining: the batch is split into groups that run their rounds on separate
    // streams, so that one group's (latency-bound) solver step overlaps another group's evaluation
    // mpc_solve_batch_async: the host side of a solve (its round loop) on a worker thread of the handle
    struct AsyncJob { int B; const double *x0, *cl; const int32_t *cl_index; double *U, *lambda, *stats; void *stream; };
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    AsyncJob job{};
    bool job_posted = false, job_running = false, job_done = false, worker_quit = false;
    int job_rc = MPC_OK;
    std::string job_err;
    int ngroups = 0; // 0 = choose from the batch size
    int groups_last = 0; // sub-batch groups of the last solve
    long long round_limit = 0; // mpc_set_round_limit: cap on the rounds / persistent-kernel trips of a solve (0 = the guard alone)
    int hw_queues = 4; // streams of this process the HIP runtime runs side by side: 5 (or more) / 4 (or fewer), measured once per
                       // process and device (probe_stream_concurrency)
    double poll_timeout_s = 300.0; // wall-clock bound of a solve's host waits (mpc_set_poll_timeout / MPC_POLL_TIMEOUT_S): the
                                   // round loop gives up when no polled window has completed for this long, the blocking waits
                                   // behind it when they have lasted this long.  A valid solve never comes near it.
    bool timed_out = false;        // the last solve ended on that bound: work may still be queued on the device
    hipEvent_t syncev = nullptr;   // bounded_sync
    hipStream_t gstream[MPC_MAX_GROUPS] = {};
    hipEvent_t gevent[MPC_MAX_GROUPS + 1] = {};
    // staging buffers for the standalone entry points
    double *stage = nullptr;
    size_t stage_bytes = 0;
};

extern "C" const char *mpc_last_error(void) { return g_err.c_str(); }
#ifndef MPC_SOURCE_SHA256
#define MPC_SOURCE_SHA256 "unknown"
#endif
extern "C" const char *mpc_source_hash(void) { return MPC_SOURCE_SHA256; }

extern "C" int mpc_default_config(mpc_config *c, int model, int N)
{
    if (!c || N < 1 || N > MPC_MAX_N || (model != MPC_MODEL_KINEMATIC && model != MPC_MODEL_PACEJKA))
        return fail(MPC_E_ARG, "mpc_default_config: bad model or horizon");
    std::memset(c, 0, sizeof(*c));
    c->model = model; c->N = N;
    c->S = 100;            // main.py:70
    c->nfe = 4;            // car_dynamics.py:136
    c->wrap_mode = MPC_WRAP_FLOOR;
    c->lbfgs_memory = N;   // controller.py:36
    c->max_iter = 1000;    // controller.py:31
    c->max_outer = 1000;   // controller.py:45
    c->hess_heuristic = 15; // controller.py:32
    c->max_no_progress = 10;
    c->Ts = 0.05;          // car_dynamics.py:93
    c->v_ref = 1.0;        // main.py:65
    const double w[6] = {0.5, 1.0, 1.0, 0.5, 0.1, 0.01}; // car_dynamics.py:230
    std::memcpy(c->cost_w, w, sizeof w);
    const double veh[22] = {9.7e-2, 4.7e-2, 5e-2, 0.09, 0.07, 8e-2, 5.5e-2, 0.1735, 18.3e-5, // main.py:84
                            0.32, 1.0,                                                      // main.py:82
                            0.268, 2.165, 3.47, 0.242, 2.38, 2.84,                          // main.py:85
                            0.266, 0.1, 0.1025, 0.1629, 0.0011};                            // main.py:86
    std::memcpy(c->veh, veh, sizeof veh);
    c->accel = 2.0; c->friction = 1.0; // dynamics.py:34-35
    c->u_lb[0] = -1.0; c->u_ub[0] = 1.0; c->u_lb[1] = -0.32; c->u_ub[1] = 0.32; // main.py:55-56
    const double off[6] = {20, 1, 1, 2, 1, 0.1}; // main.py:46-51
    std::memcpy(c->g_off, off, sizeof off);
    for (int i = 0; i < 6; i++) { c->D_lb[i] = -INFINITY; c->D_ub[i] = INFINITY; }
    c->lane_halfwidth = 0.15;
    c->alm_eps = 1e-6; c->alm_delta = 1e-4; c->Sigma0 = 1e5; // controller.py:41-43
    c->eps0 = 1.0; c->rho = 0.1; c->Delta = 10.0; c->theta = 0.1; c->M = 1e9; c->Sigma_max = 1e9;
    c->Delta_lower = 0.8; c->Sigma0_lower = 0.6; c->eps0_increase = 1.1; c->rho_increase = 2.0;
    c->max_num_initial_retries = 20; c->max_num_retries = 20; c->max_total_num_retries = 40;
    c->max_total_inner = 5000; c->max_total_evals = 0;
    c->lip_eps = 1e-6; c->lip_delta = 1e-12; c->Lgamma_factor = 0.95;
    c->L_min = 1e-5; c->L_max = 1e20; c->tau_min = 1.0 / 256; c->qub_tol = 10 * DBL_EPSILON;
    return MPC_OK;
}

extern "C" int mpc_nx(const mpc_config *c) { return c->model == MPC_MODEL_PACEJKA ? 6 : 4; }
static int stage_m(const mpc_config *c)
{
    return c->constr_mode == MPC_CONSTR_STATE_SQ ? mpc_nx(c) : c->constr_mode == MPC_CONSTR_LANE ? 1 : 0;
}
extern "C" int mpc_m(const mpc_config *c) { return stage_m(c) * c->N; }

static int make_devcfg(const mpc_config &c, DevCfg &d)
{
    if (c.N < 1 || c.N > MPC_MAX_N) return fail(MPC_E_ARG, "horizon N out of range [1, 64]");
    if (c.S < 3) return fail(MPC_E_ARG, "centerline needs S >= 3 points");
    if (c.nfe < 1 || c.nfe > 16) return fail(MPC_E_ARG, "nfe out of range [1, 16]");
    if (c.lbfgs_memory < 1 || c.lbfgs_memory > 64) return fail(MPC_E_ARG, "lbfgs_memory out of range [1, 64]");
    if (c.model != MPC_MODEL_KINEMATIC && c.model != MPC_MODEL_PACEJKA) return fail(MPC_E_ARG, "unknown model");
    if (c.constr_mode < 0 || c.constr_mode > 2) return fail(MPC_E_ARG, "unknown constr_mode");
    if (c.max_no_progress < 1) return fail(MPC_E_ARG, "max_no_progress must be >= 1");
    if (c.max_iter < 1 || c.max_outer < 1 || c.max_total_inner < 1 || c.max_total_evals < 0)
        return fail(MPC_E_ARG, "max_iter, max_outer, max_total_inner must be >= 1 and max_total_evals >= 0");
    if (c.max_num_initial_retries < 0 || c.max_num_retries < 0 || c.max_total_num_retries < 0)
        return fail(MPC_E_ARG, "retry limits must be >= 0");
    if (!(c.Ts > 0.0) || !std::isfinite(c.Ts)) return fail(MPC_E_ARG, "Ts must be positive and finite");
    // alpaqa has a separate initial-penalty path for Sigma_0 == 0; it is not restated here
    if (!(c.Sigma0 > 0.0) || !(c.Sigma_max >= c.Sigma0) || !(c.M >= 0.0))
        return fail(MPC_E_ARG, "need 0 < Sigma0 <= Sigma_max and M >= 0");
    if (!(c.L_min > 0.0) || !(c.L_min <= c.L_max)) return fail(MPC_E_ARG, "need 0 < L_min <= L_max");
    if (!(c.alm_eps > 0.0) || !(c.alm_delta > 0.0) || !(c.eps0 > 0.0))
        return fail(MPC_E_ARG, "tolerances alm_eps, alm_delta, eps0 must be positive");
    if (!(c.tau_min > 0.0) || !(c.tau_min <= 1.0)) return fail(MPC_E_ARG, "tau_min must be in (0, 1]");
    if (!(c.Lgamma_factor > 0.0) || !(c.Lgamma_factor < 1.0)) return fail(MPC_E_ARG, "Lgamma_factor must be in (0, 1)");
    for (int i = 0; i < 2; i++)
        if (!(c.u_lb[i] <= c.u_ub[i])) return fail(MPC_E_ARG, "input box: u_lb must not exceed u_ub");
    std::memset(&d, 0, sizeof d);
    d.model = c.model; d.N = c.N; d.S = c.S; d.nfe = c.nfe; d.wrap_mode = c.wrap_mode;
    d.clip_inputs = c.clip_inputs; d.constr_mode = c.constr_mode; d.sm = stage_m(&c);
    d.nx = mpc_nx(&c); d.n = 2 * c.N; d.m = d.sm * c.N; d.M = c.lbfgs_memory;
    d.max_iter = c.max_iter; d.max_outer = c.max_outer; d.hess_heuristic = c.hess_heuristic;
    d.max_no_progress = c.max_no_progress;
    d.max_num_initial_retries = c.max_num_initial_retries; d.max_num_retries = c.max_num_retries;
    d.max_total_num_retries = c.max_total_num_retries; d.max_total_inner = c.max_total_inner;
    d.max_total_evals = c.max_total_evals;
    d.no_spec = getenv("MPC_NO_SPEC") != nullptr;
    d.no_memo = getenv("MPC_NO_MEMO") != nullptr;
    d.no_la = getenv("MPC_NO_LOOKAHEAD") != nullptr;
    d.all_rows = getenv("MPC_ALL_ROWS") != nullptr;
    // (MPC_NO_CHAIN: never; which launches carry them is decided per launch: chain_min)
    // (kinematic model only by default: measured on the Pacejka model, whose rounds wait for the rollout, 668 -> 699 ms per
    // solve with them; MPC_CHAIN_MIN set explicitly turns them on for either model)
    d.chain = getenv("MPC_NO_CHAIN") == nullptr && 2 * c.N <= 64 &&
              (c.model == MPC_MODEL_KINEMATIC || getenv("MPC_CHAIN_MIN") != nullptr);
    d.h = c.Ts / c.nfe; d.v_ref = c.v_ref;
    for (int i = 0; i < 6; i++) { d.w[i] = c.cost_w[i]; d.g_off[i] = c.g_off[i]; d.D_lb[i] = c.D_lb[i]; d.D_ub[i] = c.D_ub[i]; }
    d.lf = c.veh[1]; d.lr = c.veh[2]; d.mass = c.veh[7]; d.inv_mass = 1.0 / c.veh[7]; d.inv_iz = 1.0 / c.veh[8];
    d.max_steer = c.veh[9]; d.max_drive = c.veh[10];
    d.bf = c.veh[11]; d.cf = c.veh[12]; d.df = c.veh[13]; d.br = c.veh[14]; d.cr = c.veh[15]; d.dr = c.veh[16];
    d.cm1 = c.veh[17]; d.cm2 = c.veh[18]; d.cr0 = c.veh[19]; d.cr2 = c.veh[21];
    d.accel = c.accel; d.friction = c.friction;
    for (int i = 0; i < 2; i++) { d.u_lb[i] = c.u_lb[i]; d.u_ub[i] = c.u_ub[i]; }
    d.lane_hw = c.lane_halfwidth;
    d.alm_eps = c.alm_eps; d.alm_delta = c.alm_delta; d.Sigma0 = c.Sigma0; d.eps0 = c.eps0; d.rho = c.rho;
    d.Delta = c.Delta; d.theta = c.theta; d.Mcap = c.M; d.Sigma_max = c.Sigma_max;
    d.Delta_lower = c.Delta_lower; d.Sigma0_lower = c.Sigma0_lower; d.eps0_increase = c.eps0_increase;
    d.rho_increase = c.rho_increase;
    d.lip_eps = c.lip_eps; d.lip_delta = c.lip_delta; d.Lgamma = c.Lgamma_factor; d.L_min = c.L_min;
    d.L_max = c.L_max; d.tau_min = c.tau_min; d.qub_tol = c.qub_tol;
    return MPC_OK;
}

// How many of this process's streams the HIP runtime runs side by side.  It maps streams to
// GPU_MAX_HW_QUEUES hardware queues (4 unless its environment said otherwise WHEN IT INITIALISED -- the
// variable as this process sees it now may have been set too late to count), and two streams that share a
// queue serialise: four sub-batch groups beside the caller's stream on four queues cost 259.8 ms per solve
// against 169.8 ms for three.  So the group count is decided on what is measured here, once
// per handle: a kernel that idles for a fixed time on the caller-side null stream and on the four group
// streams; side by side they take one such time, sharing a queue two.
__global__ void spin_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
static int probe_stream_concurrency_once(int device)
{
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;
    const double spin_us = 250.0;
    const long long ticks = (long long)(spin_us * 1e-6 * khz * 1e3);
    // five private non-blocking streams (not the null stream: a probe must neither wait for nor hold up the caller's
    // other streams, and must work while the caller is capturing a graph elsewhere); only they are synchronised
    hipStream_t st[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int result = 4;
    bool ok = true;
    for (int k = 0; k < 5 && ok; k++) ok = hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[0], 1LL);    // code object load, first-launch costs
        ok = hipStreamSynchronize(st[0]) == hipSuccess;
    }
    double best = 1e30;
    // three samples; when even the best of them looks like a shared queue AND like a busy device (more than three spins:
    // another handle's solve was running beside the probe), sample again a few times before settling for "four"
    for (int rep = 0; rep < 9 && ok; rep++) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 5; k++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[k], ticks);
        for (int k = 0; k < 5 && ok; k++) ok = hipStreamSynchronize(st[k]) == hipSuccess;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        best = std::min(best, us);
        if (rep >= 2 && !(best > 3.0 * spin_us)) break;
    }
    (void)hipGetLastError();
    for (int k = 0; k < 5; k++) if (st[k]) (void)hipStreamDestroy(st[k]);
    // five side by side: ~ one spin (+ launch overheads); a shared queue: two spins or more
    if (ok) result = best < 1.6 * spin_us ? 5 : 4;
    return result;
}
// measured once per (process, device): the answer is a property of the runtime's queue setup, not of the handle, and a
// handle created while another one is solving must not keep a pessimistic sample for its lifetime
static int probe_stream_concurrency(mpc_handle *h)
{
    static std::mutex mu;
    static int cached[64];
    std::lock_guard<std::mutex> lk(mu);
    const int d = h->device >= 0 && h->device < 64 ? h->device : 0;
    if (cached[d] == 0) cached[d] = probe_stream_concurrency_once(h->device);
    return cached[d];
}

extern "C" int mpc_create(const mpc_config *cfg, int device, mpc_handle **out)
{
    if (!cfg || !out) return fail(MPC_E_ARG, "mpc_create: null argument");
    {   // the configuration is checked before anything touches the device (and without one: CPU tests)
        DevCfg probe;
        const int rcv = make_devcfg(*cfg, probe);
        if (rcv) return rcv;
    }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(MPC_E_ARG, "mpc_create: no such device");
    mpc_handle *h = new mpc_handle();
    h->step_regs = getenv("MPC_STEP_REGS") != nullptr;
    if (getenv("MPC_LDS_PAIRS")) h->lds_pairs = atoi(getenv("MPC_LDS_PAIRS"));
    if (getenv("MPC_CHAIN_MIN")) h->chain_min = atoi(getenv("MPC_CHAIN_MIN"));
    h->quad_rollout = getenv("MPC_NO_QUAD") == nullptr;
    if (getenv("MPC_PAC_QUAD_MAX")) h->pac_quad_max = atoi(getenv("MPC_PAC_QUAD_MAX"));
    h->arrive_adjoint = getenv("MPC_ARRIVE") != nullptr;
    if (getenv("MPC_WIDE_MAX")) h->wide_max = atoi(getenv("MPC_WIDE_MAX"));
    if (getenv("MPC_APB")) h->apb_env = atoi(getenv("MPC_APB"));
    h->fused_eval = getenv("MPC_UNFUSED_EVAL") == nullptr;
    if (getenv("MPC_FUSED_MAX")) h->fused_max = atoi(getenv("MPC_FUSED_MAX"));
    // Pacejka model: 128.  Its persistent-kernel waves take a whole SIMD each (512 registers); the 2 237 agents that four
    // groups hand over at 1 024 requests each are more than the chip's 1 024 SIMDs hold, the later groups' waves queue and
    // the ones in flight starve the other groups' rounds, while a thin Pacejka round is no slower per evaluation than a
    // lone wave's trip (round 4, same box, alternating, 65 536 agents, bit-identical: 1 024 -> 462 ms, 512 -> 447,
    // 384 -> 426, 32 .. 256 -> 412 - 429).  Kinematic model: 1 024 as measured in round 2 (its trips are 4x shorter than
    // a thin round, two waves per SIMD).
    h->solo_max = cfg->model == MPC_MODEL_PACEJKA ? 128 : 1024;
    h->solo_all = (cfg->model == MPC_MODEL_KINEMATIC && cfg->N <= 32) ? 4096 : 1024;
    if (getenv("MPC_SOLO_MAX")) h->solo_max = h->solo_all = atoi(getenv("MPC_SOLO_MAX"));
    if (getenv("MPC_SOLO_ALL")) h->solo_all = atoi(getenv("MPC_SOLO_ALL"));
    h->nearest_mode = getenv("MPC_NEAREST_SCAN") ? 0 : getenv("MPC_NEAREST_BLOCKS") ? 1 : 2;
    h->cfg = *cfg;
    int rc = make_devcfg(*cfg, h->dc);
    if (rc) { delete h; return rc; }
    if (h->arrive_adjoint) h->dc.chain = 0;   // (the experimental K1c-inside-K1b variant leaves no gradient-slot count for chain_block)
    h->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->num_cus = cus;
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->host_counts, 512, hipHostMallocDefault);
    if (e != hipSuccess) { delete h; return fail(MPC_E_HIP, std::string("mpc_create: ") + hipGetErrorString(e)); }
    const char *p = getenv("MPC_PROFILE");
    h->profile = p && p[0] == '1';
    if (getenv("MPC_POLL_TIMEOUT_S")) { const double t = atof(getenv("MPC_POLL_TIMEOUT_S")); if (t > 0.0) h->poll_timeout_s = t; }
    const char *gq = getenv("MPC_GROUPS");
    h->ngroups = gq ? atoi(gq) : 0;
    // (MPC_HW_QUEUES overrides the measurement: experiments only)
    h->hw_queues = getenv("MPC_HW_QUEUES") ? atoi(getenv("MPC_HW_QUEUES")) : probe_stream_concurrency(h);
    *out = h;
    return MPC_OK;
}

extern "C" int mpc_destroy(mpc_handle *h)
{
    if (!h) return MPC_OK;
    if (h->worker.joinable()) {           // a solve still running finishes first
        { std::lock_guard<std::mutex> lk(h->mu); h->worker_quit = true; }
        h->cv.notify_all();
        h->worker.join();
    }
    (void)hipSetDevice(h->device);
    if (h->arena) (void)hipFree(h->arena);
    if (h->stage) (void)hipFree(h->stage);
    if (h->cl_boxes) (void)hipFree(h->cl_boxes);
    if (h->cl_gmeta) (void)hipFree(h->cl_gmeta);
    if (h->cl_gxy) (void)hipFree(h->cl_gxy);
    if (h->cl_gcells) (void)hipFree(h->cl_gcells);
    if (h->host_counts) (void)hipHostFree(h->host_counts);
    if (h->syncev) (void)hipEventDestroy(h->syncev);
    for (auto ev : h->ev_pool) (void)hipEventDestroy(ev);
    for (int g = 0; g < MPC_MAX_GROUPS; g++) if (h->gstream[g]) (void)hipStreamDestroy(h->gstream[g]);
    for (int g = 0; g <= MPC_MAX_GROUPS; g++) if (h->gevent[g]) (void)hipEventDestroy(h->gevent[g]);
    for (int b = 0; b < 2; b++) for (int g = 0; g < MPC_MAX_GROUPS; g++) if (h->pollev[b][g]) (void)hipEventDestroy(h->pollev[b][g]);
    for (int b = 0; b < 2; b++) for (int g = 0; g < MPC_MAX_GROUPS; g++) if (h->soloev[g][b]) (void)hipEventDestroy(h->soloev[g][b]);
    delete h;
    return MPC_OK;
}

// carve the workspace for up to B agents (agent-major rows; caller buffers are used in place)
static int reserve(mpc_handle *h, int B)
{
    const DevCfg &c = h->dc;
    const int Bp = (B + 63) & ~63;
    if (Bp <= h->Bp_alloc) { h->ws.Bp = h->Bp_alloc; h->ws.B = B; return MPC_OK; }
    HIPCHK(hipSetDevice(h->device));
    if (h->arena) { HIPCHK(hipFree(h->arena)); h->arena = nullptr; h->Bp_alloc = 0; }
    const size_t n = c.n, m = c.m ? c.m : 1, M = c.M, nx = c.nx, N = c.N;
    // a round holds at most two requests per agent (cost + speculative gradient): 2 Bp slots
    const size_t JS = nx * (nx + 1) + 2, St = 2 * (size_t)Bp + 64 * (MPC_MAX_GROUPS + 1);
    const size_t nd = 8 * n + 2 * M * n + 7 * m + REC;          // agent-major doubles per agent
    const size_t nscr = (N + 1) * nx + 2 * N + N + N * JS;       // K1 scratch doubles per slot
    const size_t ni = 4;                                         // list ints per agent
    const size_t bytes = (nd * 8 + ni * 4) * (size_t)Bp + nscr * 8 * St + 4 * St + 8 * 4 * MPC_MAX_GROUPS + 256 + 64 +
                         4 * (St / 64 + 16) + 256;
    char *base = nullptr;
    hipError_t e = hipMalloc((void **)&base, bytes);
    if (e != hipSuccess) return fail(MPC_E_ALLOC, "workspace hipMalloc failed: " + std::string(hipGetErrorString(e)));
    h->arena = base; h->arena_bytes = bytes; h->Bp_alloc = Bp;
    Workspace &w = h->ws;
    double *dp = (double *)base;
    auto takeD = [&](size_t cnt) { double *r = dp; dp += cnt * (size_t)Bp; return r; };
    w.xk = takeD(n); w.gk = takeD(n); w.q = takeD(n); w.xn = takeD(n); w.xe = takeD(n); w.ge = takeD(n);
    w.xe2 = takeD(n); w.ge2 = takeD(n);
    w.S = takeD(M * n); w.Y = takeD(M * n);
    w.Sig = takeD(m); w.Sig_old = takeD(m); w.e1 = takeD(m); w.e2 = takeD(m);
    w.yhx = takeD(m); w.yhxn = takeD(m); w.yhe = takeD(m);
    w.rec = takeD(REC);
    auto takeS = [&](size_t cnt) { double *r = dp; dp += cnt * St; return r; };
    w.trajx = takeS((N + 1) * nx); w.useq = takeS(2 * N); w.stage_L = takeS(N); w.jac = takeS(N * JS);
    int *ip = (int *)dp;
    auto takeI = [&](size_t cnt) { int *r = ip; ip += cnt * (size_t)Bp; return r; };
    w.lists = takeI(4);
    w.agent_of = ip; ip += St;
    w.counts = ip; // 8 ints per group
    w.totals = (unsigned long long *)(ip + 8 * MPC_MAX_GROUPS);
    w.solo_ctr = (int *)(w.totals + 16); // [group][claim counter, list length]
    h->arrive_buf = w.solo_ctr + 2 * MPC_MAX_GROUPS + 32; // [St / 64 + 16]: stage blocks done per slot block
    w.arrive = nullptr;
    w.Bp = Bp; w.B = B; w.St = (int)St; w.Ls = Bp;
    w.ws_xe = w.xe; w.ws_ge = w.ge; w.ws_yhe = w.yhe; w.ws_Sig = w.Sig;
    HIPCHK(hipMemset(base, 0, bytes));
    return MPC_OK;
}

static int reserve_stage(mpc_handle *h, size_t bytes)
{
    if (bytes <= h->stage_bytes) return MPC_OK;
    if (h->stage) { HIPCHK(hipFree(h->stage)); h->stage = nullptr; h->stage_bytes = 0; }
    hipError_t e = hipMalloc((void **)&h->stage, bytes);
    if (e != hipSuccess) return fail(MPC_E_ALLOC, "staging hipMalloc failed");
    h->stage_bytes = bytes;
    return MPC_OK;
}

static inline dim3 grid_for(int B, int block) { return dim3((unsigned)((B + block - 1) / block)); }

template <int MODEL>
static bool launch_eval_t(mpc_handle *h, const Workspace &w, hipStream_t s, const int *lists, const int *counts,
                          int nG, int nC, hipEvent_t eva = nullptr, hipEvent_t evb = nullptr, int slot_bound = -1,
                          int *desc = nullptr)
{
    const DevCfg &c = h->dc;
    const bool shared = w.cl_index == nullptr;
    // list mode: the grid covers the most requests the round can hold -- every agent on both lists
    // (cost + speculative gradient), or the caller's tighter bound (blocks beyond the lists exit at once,
    // but late in a solve dispatching thousands of them costs more than the work)
    int nblk = counts ? 2 * (w.Bp / 64) : ((nG + 63) / 64 + (nC + 63) / 64);
    if (counts && slot_bound >= 0) nblk = std::min(nblk, (slot_bound + 126) / 64 + 1);
    if (nblk == 0) return true;
    const size_t lds = sizeof(double) * 64 * (size_t)(c.n + 1) + 64 * sizeof(int);
    bool wide = false;
    if constexpr (MODEL == KIN) {
        // few requests (late rounds of a solve, small batches): one wave per request, see rollout_wide_kernel
        wide = counts && slot_bound >= 0 && slot_bound <= h->wide_max && c.nfe == 4 && c.N <= 64;
        if (wide)
            hipLaunchKernelGGL(rollout_wide_kernel, dim3((unsigned)(nblk * 16)), dim3(256), 0, s, c, w, lists, counts);
    }
    bool quad = false;
    if constexpr (MODEL == KIN) {
        // two lanes per request (rollout_pair_kernel); the wave-per-request kernel keeps the rounds with few requests
        quad = !wide && h->quad_rollout && c.nfe == 4 && c.N <= 64;   // (its wave-wide redo of a request: kin_wide_rollout)
        if (quad)
            hipLaunchKernelGGL(rollout_pair_kernel, dim3((unsigned)(nblk * 2)), dim3(64),
                               sizeof(double) * 32 * (size_t)(c.n + 1), s, c, w, lists, counts, nG, nC);
    }
    if constexpr (MODEL == PAC) {
        // four lanes per request while a round holds few requests (a shorter chain per request where waves are alone on
        // their SIMDs); one thread per request -- 2.2 times fewer instructions in all -- once the launch fills the chip
        // (pac_quad_max: requests bound up to which the four-lane kernel runs; since the lost stages are parked no
        // request drags its wave, and the full rounds are bound by what they execute: profiles/r03_experiments.txt 34)
        quad = h->quad_rollout && !(counts && slot_bound >= 0 && slot_bound > h->pac_quad_max);
        if (quad)
            hipLaunchKernelGGL(rollout_quad_kernel, dim3((unsigned)(nblk * 4)), dim3(64),
                               sizeof(double) * 16 * (size_t)(c.n + 1), s, c, w, lists, counts, nG, nC);
    }
    if (!wide && !quad)
        hipLaunchKernelGGL((rollout_kernel<MODEL>), dim3((unsigned)nblk), dim3(64), lds, s, c, w, lists, counts, nG, nC);
    if (eva) (void)hipEventRecord(eva, s);
    // (kinematic model only: the Pacejka stage needs more registers than the fused kernel leaves it)
    if (MODEL == KIN && h->fused_eval && (!counts || (slot_bound >= 0 && slot_bound <= h->fused_max))) {
        // K1b + K1c in one launch, stage records through LDS (see stage_adjoint_kernel)
        constexpr int BLK = FusedBlk<MODEL>::BLK, JS = JacRec<MODEL>::SIZE;
        const int spb = BLK / c.N;
        const int gb = (nblk * 64 + spb - 1) / spb;
        const size_t flds = sizeof(double) * (size_t)(JS + 1) * c.N * spb;
        if (shared)
            hipLaunchKernelGGL((stage_adjoint_kernel<MODEL, true>), dim3((unsigned)gb), dim3(BLK), flds, s, c, w, counts, nG, nC, desc);
        else
            hipLaunchKernelGGL((stage_adjoint_kernel<MODEL, false>), dim3((unsigned)gb), dim3(BLK), flds, s, c, w, counts, nG, nC, desc);
        if (evb) (void)hipEventRecord(evb, s);
        return true;
    }
    // (tried: nblk rounded up to a multiple of 8, which puts every stage block of slot block sb and its adjoint
    // block on XCD sb % 8 so that K1c could read records from the L2 they were written to -- no change: the 13 MB
    // of records per XCD and launch pass through a 4 MB L2 long before K1c starts)
    const size_t xy_lds = (shared && w.near.gmeta && c.S <= GRID_LDS_MAX_S) ? sizeof(double) * 2 * (size_t)c.S : 0;
    if (shared)
        hipLaunchKernelGGL((stage_kernel<MODEL, true>), dim3((unsigned)(nblk * c.N)), dim3(64), xy_lds, s, c, w, counts, nG, nC, nblk);
    else
        hipLaunchKernelGGL((stage_kernel<MODEL, false>), dim3((unsigned)(nblk * c.N)), dim3(64), 0, s, c, w, counts, nG, nC, nblk);
    if (evb) (void)hipEventRecord(evb, s);
    if (w.arrive) return true;               // K1c ran inside K1b (last-arriving stage block)
    hipLaunchKernelGGL((adjoint_kernel<MODEL>), dim3((unsigned)nblk), dim3(64), 0, s, c, w, counts, nG, nC, desc);
    return false;
}
// returns true when K1b and K1c ran as one launch
static bool launch_eval(mpc_handle *h, const Workspace &w, hipStream_t s, const int *lists, const int *counts,
                        int nG, int nC, hipEvent_t eva = nullptr, hipEvent_t evb = nullptr, int slot_bound = -1,
                        int *desc = nullptr)
{
    if (h->dc.model == PAC) return launch_eval_t<PAC>(h, w, s, lists, counts, nG, nC, eva, evb, slot_bound, desc);
    return launch_eval_t<KIN>(h, w, s, lists, counts, nG, nC, eva, evb, slot_bound, desc);
}

// Every entry point that touches the handle's tables, workspace or streams goes through here.  While an
// asynchronous solve is posted, running or waiting to be collected (mpc_solve_batch_async .. mpc_solve_wait)
// the worker thread owns the handle: anything else is refused BEFORE it touches the handle (a second
// mpc_centerline_blocks would free or overwrite the search tables under the running solve's kernels).
static int refuse_if_busy(mpc_handle *h, const char *who)
{
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->job_posted || h->job_running || h->job_done)
        return fail(MPC_E_ARG, std::string(who) + ": a solve of this handle is in flight (mpc_solve_wait first)");
    return MPC_OK;
}
static int check_common(mpc_handle *h, int B, const char *who, bool from_worker = false)
{
    if (!h) return fail(MPC_E_ARG, std::string(who) + ": null handle");
    if (B < 0) return fail(MPC_E_ARG, std::string(who) + ": negative batch");
    if (!from_worker) { const int rb = refuse_if_busy(h, who); if (rb) return rb; }
    HIPCHK(hipSetDevice(h->device));
    return MPC_OK;
}

// the search tables to use with centerline table `cl` (all null: none prepared for it, or switched off)
static NearTab near_for(const mpc_handle *h, const double *cl)
{
    NearTab nt = {nullptr, nullptr, nullptr, nullptr};
    if (!h->cl_boxes_for || h->cl_boxes_for != cl) return nt;
    if (h->nearest_mode == 1 && h->cl_boxes) nt.boxes = h->cl_boxes;
    if (h->nearest_mode == 2 && h->cl_gmeta && h->cl_grid_ok) { nt.gmeta = h->cl_gmeta; nt.gcells = h->cl_gcells; nt.gxy = h->cl_gxy; }
    return nt;
}

extern "C" int mpc_centerline_blocks(mpc_handle *h, const double *cl, int C, void *stream)
{
    int rc = check_common(h, C, "mpc_centerline_blocks"); if (rc) return rc;
    h->cl_boxes_for = nullptr; h->cl_boxes_rows = 0;
    if (C == 0 || !cl) return MPC_OK;
    const DevCfg &c = h->dc;
    const int NB = (c.S - 1 + NEAR_BLK - 1) / NEAR_BLK;
    const bool blocks = NB <= 64;         // the block search keeps one bit per block: longer tables do without it
    const size_t bytes = sizeof(double) * 4 * (size_t)NB * (size_t)C;
    if (blocks && bytes > h->cl_boxes_bytes) {
        if (h->cl_boxes) { HIPCHK(hipFree(h->cl_boxes)); h->cl_boxes = nullptr; h->cl_boxes_bytes = 0; }
        if (hipMalloc((void **)&h->cl_boxes, bytes) != hipSuccess) return fail(MPC_E_ALLOC, "centerline block table hipMalloc failed");
        h->cl_boxes_bytes = bytes;
    }
    if (!blocks && h->cl_boxes) { HIPCHK(hipFree(h->cl_boxes)); h->cl_boxes = nullptr; h->cl_boxes_bytes = 0; }
    // the grid costs 256 KB per centerline row: a table with one row per agent keeps the full scan
    const bool grid = C <= MPC_GRID_MAX_ROWS;
    h->cl_grid_ok = false;
    if (grid && C > h->cl_grid_cap) {
        if (h->cl_gmeta) { HIPCHK(hipFree(h->cl_gmeta)); h->cl_gmeta = nullptr; }
        if (h->cl_gxy) { HIPCHK(hipFree(h->cl_gxy)); h->cl_gxy = nullptr; }
        if (h->cl_gcells) { HIPCHK(hipFree(h->cl_gcells)); h->cl_gcells = nullptr; }
        h->cl_grid_cap = 0;
        if (hipMalloc((void **)&h->cl_gmeta, sizeof(double) * GRID_META * (size_t)C) != hipSuccess ||
            hipMalloc((void **)&h->cl_gxy, sizeof(double) * 2 * (size_t)c.S * (size_t)C) != hipSuccess ||
            hipMalloc((void **)&h->cl_gcells, sizeof(unsigned) * (size_t)GRID_CELLS * (size_t)C) != hipSuccess)
            return fail(MPC_E_ALLOC, "centerline grid hipMalloc failed");
        h->cl_grid_cap = C;
    }
    hipStream_t s = (hipStream_t)stream;
    if (blocks) hipLaunchKernelGGL(cl_blocks_kernel, grid_for(C * NB, 256), dim3(256), 0, s, c, cl, C, h->cl_boxes);
    if (grid) {
        hipLaunchKernelGGL(cl_grid_meta_kernel, dim3((unsigned)C), dim3(64), 0, s, c, cl, C, h->cl_gmeta, h->cl_gxy);
        hipLaunchKernelGGL(cl_grid_cells_kernel, dim3(GRID_CELLS / 256, C), dim3(256), 0, s, c, cl, C, h->cl_gmeta, h->cl_gcells);
    }
    HIPCHK(hipGetLastError());
    h->cl_grid_ok = grid;
    h->cl_boxes_for = cl; h->cl_boxes_rows = C;
    return MPC_OK;
}

extern "C" int mpc_rhs(mpc_handle *h, int B, const double *x, const double *u, double *dx, void *stream)
{
    int rc = check_common(h, B, "mpc_rhs"); if (rc) return rc;
    if (B == 0) return MPC_OK;
    if (!x || !u || !dx) return fail(MPC_E_ARG, "mpc_rhs: null buffer");
    hipStream_t s = (hipStream_t)stream;
    if (h->dc.model == PAC) hipLaunchKernelGGL(rhs_kernel<PAC>, grid_for(B, 64), dim3(64), 0, s, h->dc, B, x, u, dx);
    else hipLaunchKernelGGL(rhs_kernel<KIN>, grid_for(B, 64), dim3(64), 0, s, h->dc, B, x, u, dx);
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_rollout(mpc_handle *h, int B, int Nsim, const double *x0, const double *U, double *X,
                           void *stream)
{
    int rc = check_common(h, B, "mpc_rollout"); if (rc) return rc;
    if (B == 0 || Nsim == 0) return MPC_OK;
    if (Nsim < 0 || !x0 || !U || !X) return fail(MPC_E_ARG, "mpc_rollout: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (h->dc.model == PAC) hipLaunchKernelGGL(simulate_kernel<PAC>, grid_for(B, 64), dim3(64), 0, s, h->dc, B, Nsim, x0, U, X);
    else hipLaunchKernelGGL(simulate_kernel<KIN>, grid_for(B, 64), dim3(64), 0, s, h->dc, B, Nsim, x0, U, X);
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_stage_errors(mpc_handle *h, int B, const double *pose, const double *cl,
                                const int32_t *cl_index, double *err, int32_t *idx, void *stream)
{
    int rc = check_common(h, B, "mpc_stage_errors"); if (rc) return rc;
    if (B == 0) return MPC_OK;
    if (!pose || !cl || !err) return fail(MPC_E_ARG, "mpc_stage_errors: null buffer");
    hipLaunchKernelGGL(errors_kernel, grid_for(B, 64), dim3(64), 0, (hipStream_t)stream, h->dc, B, pose, cl,
                       cl_index, near_for(h, cl), err, idx);
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_lane_payoff(mpc_handle *h, int B, int K, const double *params15, const double *ego,
                               const double *cars, const int32_t *ncars, double *out, void *stream)
{
    int rc = check_common(h, B, "mpc_lane_payoff"); if (rc) return rc;
    if (B == 0) return MPC_OK;
    if (K < 0 || K > 62 || !params15 || !ego || (K > 0 && !cars) || !ncars || !out)
        return fail(MPC_E_ARG, "mpc_lane_payoff: bad argument");
    LaneParams p;
    std::memcpy(&p, params15, sizeof p); // host array of 15 doubles
    hipLaunchKernelGGL(lane_payoff_kernel, grid_for(B, 64), dim3(64), 0, (hipStream_t)stream, p, B, K, ego, cars,
                       ncars, out);
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_math_probe(mpc_handle *h, int n, int op, const double *a, const double *b, double *out,
                              void *stream)
{
    int rc = check_common(h, n, "mpc_math_probe"); if (rc) return rc;
    if (n == 0) return MPC_OK;
    if (!a || !out || (op == 3 && !b)) return fail(MPC_E_ARG, "mpc_math_probe: null buffer");
    hipLaunchKernelGGL(math_probe_kernel, grid_for(n, 256), dim3(256), 0, (hipStream_t)stream, n, op, a, b, out);
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_stage_cost(mpc_handle *h, int B, const double *x, const double *u, const double *cl,
                              const int32_t *cl_index, double *out, void *stream)
{
    int rc = check_common(h, B, "mpc_stage_cost"); if (rc) return rc;
    if (B == 0) return MPC_OK;
    if (!x || !u || !cl || !out) return fail(MPC_E_ARG, "mpc_stage_cost: null buffer");
    hipStream_t s = (hipStream_t)stream;
    if (h->dc.model == PAC) hipLaunchKernelGGL(stage_cost_kernel<PAC>, grid_for(B, 64), dim3(64), 0, s, h->dc, B, x, u, cl, cl_index, out);
    else hipLaunchKernelGGL(stage_cost_kernel<KIN>, grid_for(B, 64), dim3(64), 0, s, h->dc, B, x, u, cl, cl_index, out);
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

static int eval_cost_grad(mpc_handle *h, int B, const double *x0, const double *cl, const int32_t *cl_index,
                          const double *U, const double *y, const double *Sigma, double *psi, double *grad,
                          double *yhat, void *stream, bool wave_path);
extern "C" int mpc_eval_cost_grad(mpc_handle *h, int B, const double *x0, const double *cl,
                                  const int32_t *cl_index, const double *U, const double *y,
                                  const double *Sigma, double *psi, double *grad, double *yhat, void *stream)
{
    return eval_cost_grad(h, B, x0, cl, cl_index, U, y, Sigma, psi, grad, yhat, stream, false);
}
extern "C" int mpc_eval_cost_grad_wave(mpc_handle *h, int B, const double *x0, const double *cl,
                                       const int32_t *cl_index, const double *U, const double *y,
                                       const double *Sigma, double *psi, double *grad, double *yhat, void *stream)
{
    return eval_cost_grad(h, B, x0, cl, cl_index, U, y, Sigma, psi, grad, yhat, stream, true);
}
static int eval_cost_grad(mpc_handle *h, int B, const double *x0, const double *cl, const int32_t *cl_index,
                          const double *U, const double *y, const double *Sigma, double *psi, double *grad,
                          double *yhat, void *stream, bool wave_path)
{
    int rc = check_common(h, B, "mpc_eval_cost_grad"); if (rc) return rc;
    if (B == 0) return MPC_OK;
    if (!x0 || !cl || !U || !psi) return fail(MPC_E_ARG, "mpc_eval_cost_grad: null buffer");
    const DevCfg &c = h->dc;
    if (c.m && (!y || !Sigma)) return fail(MPC_E_ARG, "mpc_eval_cost_grad: y and Sigma are required when m > 0");
    rc = reserve(h, B); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // direct mode: the kernel reads and writes the caller's agent-major buffers in place
    Workspace w = h->ws;
    w.cl = cl; w.cl_index = cl_index; w.x0 = x0; w.near = near_for(h, cl);
    w.arrive = h->arrive_adjoint ? h->arrive_buf : nullptr;
    if (w.arrive) HIPCHK(hipMemsetAsync(h->arrive_buf, 0, sizeof(int) * (size_t)(h->ws.St / 64 + 16), s));
    w.xe = const_cast<double *>(U); w.ge = grad ? grad : h->ws.ws_ge;
    w.y = const_cast<double *>(y); w.Sig = c.m ? const_cast<double *>(Sigma) : h->ws.ws_Sig;
    w.yhe = (yhat && c.m) ? yhat : h->ws.ws_yhe;
    w.psi_direct = psi;
    Workspace saved = h->ws;
    h->ws = w;
    if (wave_path) {
        // one wave per agent, the evaluation as the persistent kernel runs it (mpc_solo.hpp)
        if (c.model == PAC) {
            const size_t lds = sizeof(double) * solo_lds_doubles<PAC>(c.nfe, c.N, c.n, c.M, false);
            hipLaunchKernelGGL(solo_eval_kernel<PAC>, dim3((unsigned)B), dim3(64), lds, s, c, h->ws, grad ? 1 : 0);
        } else {
            const size_t lds = sizeof(double) * solo_lds_doubles<KIN>(c.nfe, c.N, c.n, c.M, false);
            hipLaunchKernelGGL(solo_eval_kernel<KIN>, dim3((unsigned)B), dim3(64), lds, s, c, h->ws, grad ? 1 : 0);
        }
    } else
        launch_eval(h, h->ws, s, nullptr, nullptr, grad ? B : 0, grad ? 0 : B);
    h->ws = saved;
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

extern "C" int mpc_prox_step(mpc_handle *h, int B, const double *x, const double *grad, const double *gamma,
                             double *xhat, double *p, double *out, void *stream)
{
    int rc = check_common(h, B, "mpc_prox_step"); if (rc) return rc;
    if (B == 0) return MPC_OK;
    if (!x || !grad || !gamma || !out) return fail(MPC_E_ARG, "mpc_prox_step: null buffer");
    hipLaunchKernelGGL(prox_kernel, grid_for(B, 64), dim3(64), 0, (hipStream_t)stream, h->dc, B, x, grad, gamma,
                       xhat, p, out);
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

template <int NE, int MC>
static void launch_lbfgs_harness(const DevCfg &c, hipStream_t s, int B, const double *S, const double *Y,
                                 const int32_t *idx, const int32_t *full, const double *mask, double *q,
                                 int32_t *ok, unsigned long long *rows)
{
    hipLaunchKernelGGL((lbfgs_apply_kernel<NE, MC>), dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, c, B, S, Y,
                       idx, full, mask, q, ok, rows);
}

extern "C" int mpc_lbfgs_apply(mpc_handle *h, int B, const double *S, const double *Y, const int32_t *idx,
                               const int32_t *full, const double *mask, double *q, int32_t *ok, void *stream)
{
    int rc = check_common(h, B, "mpc_lbfgs_apply"); if (rc) return rc;
    if (B == 0) return MPC_OK;
    if (!S || !Y || !idx || !full || !mask || !q || !ok) return fail(MPC_E_ARG, "mpc_lbfgs_apply: null buffer");
    rc = reserve(h, B); if (rc) return rc;
    const DevCfg &c = h->dc;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *rows = h->ws.totals + 3;
    if (c.n <= 64) {
        if (c.M <= 20) launch_lbfgs_harness<1, 20>(c, s, B, S, Y, idx, full, mask, q, ok, rows);
        else launch_lbfgs_harness<1, 0>(c, s, B, S, Y, idx, full, mask, q, ok, rows);
    } else launch_lbfgs_harness<2, 0>(c, s, B, S, Y, idx, full, mask, q, ok, rows);
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

static hipEvent_t get_event(mpc_handle *h, size_t i)
{
    while (h->ev_pool.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[i];
}

template <int NE, int MC>
static void launch_step_t(mpc_handle *h, const Workspace &w, hipStream_t s, int *lists, int *counts, int *counts_next,
                          int slot_bound, int par)
{
    // LDS copy of an agent's L-BFGS history (MC < 0): the ring slots 0 .. P - 1.  The unconstrained variant runs four
    // waves per SIMD (128 registers), so P is what lets four workgroups share a CU's 160 KB -- 15 pairs at n = 40,
    // where an application reads 5.4 on average; a longer history reads its remaining slots from global memory
    int P = h->dc.M;
    if (MC < 0 && NE == 1 && h->dc.m == 0) {
        const size_t per_pair = (size_t)STEP_WAVES * 2 * h->dc.n * sizeof(double);
        const int fit = (int)((160 * 1024 / 4 - 512) / per_pair);
        P = std::max(1, std::min(P, fit));
    }
    if (MC < 0 && NE == 2) {
        // two elements per lane (n > 64; BASELINE config 3: N = 40, n = 80, M = 40), compiled for three waves per SIMD => three
        // workgroups per CU; the LDS copy holds what fits a third of a CU's LDS -- 10 pairs of 2 x 640 B per wave at n = 80, where an
        // application reads 11.7 pairs on average -- the rest of a long history comes from global memory.
        // Round 4: this replaces the global-memory two-loop (MC = 0), which read every pair twice: 574 GB per solve of
        // config 3 at full size.
        const size_t per_pair = (size_t)STEP_WAVES * 2 * h->dc.n * sizeof(double);
        const int fit = (int)((160 * 1024 / 3 - 1024) / per_pair);
        P = std::max(1, std::min(P, fit));
    }
    if (h->lds_pairs > 0) P = std::max(1, std::min(h->dc.M, h->lds_pairs));   // MPC_LDS_PAIRS: experiments, tests
    size_t lds = MC < 0 ? (size_t)STEP_WAVES * 2 * P * h->dc.n * sizeof(double) : 0;
    // thread-per-agent blocks for the agents that wait in PH_W_LS_G (chain_block): one per 64 gradient slots the
    // finished round can have held (the same bound that sizes the K1 grids)
    // ... only while the round is a full one: the thread-per-agent chain is ~15 us long whatever the count, which a
    // step launch of > 100 us hides and a thin round's does not (chain_min: requests bound from which they are used)
    int nchain = 0;
    DevCfg dcl = h->dc;
    dcl.chain = h->dc.chain && slot_bound >= h->chain_min;
    if (NE == 1 && dcl.chain) {
        nchain = w.Bp / 64;
        if (slot_bound >= 0) nchain = std::min(nchain, (slot_bound + 126) / 64 + 1);
        lds = std::max(lds, sizeof(double) * CHAIN_SLOTS * (size_t)(h->dc.n + 1) + sizeof(int) * CHAIN_SLOTS);
    }
    // agents per workgroup: 16 per wave fills the chip from ~50 k agents; smaller batches trade
    // throughput for latency (a wave walks its agents serially)
    const int apb_env = h->apb_env;
    const int apb = apb_env == 64 || apb_env == 32 || apb_env == 16 || apb_env == 8 || apb_env == 4 ? apb_env
                  : w.B >= 16384 ? 64 : w.B >= 6144 ? 16 : 4; // measured: B = 1 Ki, 4 Ki -> 4; 8 Ki -> 16; 21 Ki -> 64
    const int nstep = (w.B + apb - 1) / apb;
    if (h->dc.m == 0)
        hipLaunchKernelGGL((step_kernel<NE, MC, false>), dim3((unsigned)(nstep + nchain)), dim3(64 * STEP_WAVES), lds, s,
                           dcl, w, lists, counts, counts_next, apb, nstep, par, P);
    else
        hipLaunchKernelGGL((step_kernel<NE, MC, true>), dim3((unsigned)(nstep + nchain)), dim3(64 * STEP_WAVES), lds, s,
                           dcl, w, lists, counts, counts_next, apb, nstep, par, P);
}
static void launch_step(mpc_handle *h, const Workspace &w, hipStream_t s, int *lists, int *counts, int *counts_next,
                        int slot_bound, int par)
{
    const DevCfg &c = h->dc;
    if (c.n <= 64) { // one element per lane; history rows cached in registers up to M = 20
        // history of one agent in LDS (12.5 KiB per wave at M n = 800: three workgroups per CU)
        if (!h->step_regs && c.M * c.n <= 800) launch_step_t<1, -1>(h, w, s, lists, counts, counts_next, slot_bound, par);
        else if (c.M <= 20) launch_step_t<1, 20>(h, w, s, lists, counts, counts_next, slot_bound, par);
        else launch_step_t<1, 0>(h, w, s, lists, counts, counts_next, slot_bound, par);
    } else if (!h->step_regs) launch_step_t<2, -1>(h, w, s, lists, counts, counts_next, slot_bound, par);
    else launch_step_t<2, 0>(h, w, s, lists, counts, counts_next, slot_bound, par);   // MPC_STEP_REGS: the global-memory two-loop
}

// The persistent wave-per-agent kernel for the agents of view `v` that are still running (`listed`:
// a list of them is built first; otherwise every agent of the view is claimed).  `bound` = an upper
// bound on the number of agents it will find.
template <int MODEL, int NE, int MC>
static void launch_solo_t(mpc_handle *h, const Workspace &v, hipStream_t s, int *ctr, bool listed, int bound,
                          long long max_trips)
{
    const DevCfg &c = h->dc;
    int *list = listed ? v.lists : nullptr;   // the round lists are free once the group leaves the rounds
    if (listed)
        hipLaunchKernelGGL(solo_list_kernel, dim3((unsigned)((v.B + 255) / 256)), dim3(256), 0, s, v, list, ctr);
    const bool la = NE == 1 && solo_lookahead(MODEL, c.nfe, c.N, c.m, c.no_la);
    const size_t lds = sizeof(double) * SOLO_WAVES * solo_lds_doubles<MODEL>(c.nfe, c.N, c.n, c.M, MC < 0, la);
    int nblk = (bound + SOLO_WAVES - 1) / SOLO_WAVES;
    nblk = std::max(1, std::min(nblk, 4 * SoloOcc<MODEL, MC>::WPS * h->num_cus)); // what is resident (registers); the rest queues
    if constexpr (MODEL == PAC && NE == 1) {
        if (la) {
            hipLaunchKernelGGL((solo_kernel<MODEL, NE, MC, true>), dim3((unsigned)nblk), dim3(64 * SOLO_WAVES), lds, s, c, v, list,
                               ctr, max_trips);
            return;
        }
    }
    hipLaunchKernelGGL((solo_kernel<MODEL, NE, MC>), dim3((unsigned)nblk), dim3(64 * SOLO_WAVES), lds, s, c, v, list,
                       ctr, max_trips);
}
template <int MODEL>
static void launch_solo_m(mpc_handle *h, const Workspace &v, hipStream_t s, int *ctr, bool listed, int bound,
                          long long max_trips)
{
    const DevCfg &c = h->dc;
    if (c.n <= 64) { // the same variant choice as launch_step: results do not depend on it
        if (!h->step_regs && c.M * c.n <= 800) launch_solo_t<MODEL, 1, -1>(h, v, s, ctr, listed, bound, max_trips);
        else if (c.M <= 20) launch_solo_t<MODEL, 1, 20>(h, v, s, ctr, listed, bound, max_trips);
        else launch_solo_t<MODEL, 1, 0>(h, v, s, ctr, listed, bound, max_trips);
    } else launch_solo_t<MODEL, 2, 0>(h, v, s, ctr, listed, bound, max_trips);
}
static void launch_solo(mpc_handle *h, const Workspace &v, hipStream_t s, int *ctr, bool listed, int bound,
                        long long max_trips)
{
    if (h->dc.model == PAC) launch_solo_m<PAC>(h, v, s, ctr, listed, bound, max_trips);
    else launch_solo_m<KIN>(h, v, s, ctr, listed, bound, max_trips);
}
static bool solo_fits(const mpc_handle *h)
{
    const DevCfg &c = h->dc;
    const bool hist = c.n <= 64 && !h->step_regs && c.M * c.n <= 800;
    const bool la = c.n <= 64 && solo_lookahead(c.model, c.nfe, c.N, c.m, c.no_la);
    const size_t per = c.model == PAC ? solo_lds_doubles<PAC>(c.nfe, c.N, c.n, c.M, hist, la)
                                      : solo_lds_doubles<KIN>(c.nfe, c.N, c.n, c.M, hist);
    return c.N <= 64 && per * SOLO_WAVES * sizeof(double) <= 64 * 1024;
}

// hipStreamSynchronize with the handle's wall-clock bound: an event behind everything queued on `s`, polled in naps.
// Returns MPC_OK, or MPC_E_HIP with h->timed_out set when the bound expired (work is then still queued).
static int bounded_sync(mpc_handle *h, hipStream_t s, const char *what)
{
    if (!h->syncev) HIPCHK(hipEventCreateWithFlags(&h->syncev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->syncev, s));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(h->syncev);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail(MPC_E_HIP, std::string(what) + ": " + hipGetErrorString(q));
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > h->poll_timeout_s) {
            h->timed_out = true;
            return fail(MPC_E_HIP, std::string(what) + ": wall-clock bound of " + std::to_string(h->poll_timeout_s) +
                                   " s expired while waiting for the device (mpc_set_poll_timeout); work is still queued");
        }
        if (waited < 100e-6) __builtin_ia32_pause();
        else std::this_thread::sleep_for(std::chrono::microseconds(waited < 5e-3 ? 20 : 200));
    }
    (void)hipGetLastError();   // (the queries that said "not ready" left that as the thread's last error)
    return MPC_OK;
}

// a view of the workspace restricted to agents [lo, hi): local agent ids, own lists / scratch
static Workspace group_view(const Workspace &w, const DevCfg &c, int g, int lo, int hi)
{
    Workspace v = w;
    const size_t n = c.n, m = c.m, M = c.M;
    v.x0 = w.x0 + (size_t)lo * c.nx; v.xo = w.xo + (size_t)lo * n;
    v.xk = w.xk + (size_t)lo * n; v.gk = w.gk + (size_t)lo * n; v.q = w.q + (size_t)lo * n;
    v.xn = w.xn + (size_t)lo * n; v.xe = w.xe + (size_t)lo * n; v.ge = w.ge + (size_t)lo * n;
    v.xe2 = w.xe2 + (size_t)lo * n; v.ge2 = w.ge2 + (size_t)lo * n;
    v.S = w.S + (size_t)lo * M * n; v.Y = w.Y + (size_t)lo * M * n;
    if (w.y) v.y = w.y + (size_t)lo * m;
    v.Sig = w.Sig + (size_t)lo * m; v.Sig_old = w.Sig_old + (size_t)lo * m; v.e1 = w.e1 + (size_t)lo * m;
    v.e2 = w.e2 + (size_t)lo * m; v.yhx = w.yhx + (size_t)lo * m; v.yhxn = w.yhxn + (size_t)lo * m;
    v.yhe = w.yhe + (size_t)lo * m;
    v.rec = w.rec + (size_t)lo * REC;
    if (w.cl_index) v.cl_index = w.cl_index + lo;
    const size_t soff = 2 * (size_t)lo + 64 * (size_t)g; // disjoint slot intervals inside the shared scratch
    v.trajx = w.trajx + soff; v.useq = w.useq + soff; v.stage_L = w.stage_L + soff; v.jac = w.jac + soff;
    v.agent_of = w.agent_of + soff;
    if (w.arrive) v.arrive = w.arrive + soff / 64;
    v.lists = w.lists + lo;
    v.counts = w.counts + 8 * g;
    v.B = hi - lo; v.Bp = (v.B + 63) & ~63;
    return v;
}

// the solve proper; x0 / U / lambda are the caller's buffers, used in place
static int run_solver_rounds(mpc_handle *h, hipStream_t s);
static int run_solver(mpc_handle *h, hipStream_t s)
{
    h->timed_out = false;
    const int rc = run_solver_rounds(h, s);
    // On any failure rounds may still be queued on the sub-batch streams (non-blocking streams: a
    // wait on `s` does not cover them) and they write into the caller's U / lambda and the arena:
    // nothing is handed back to the caller before the device has drained -- EXCEPT after the wall-clock
    // bound: the device is not answering, a blocking wait would be the hang the bound exists to end.  The
    // caller gets MPC_E_HIP and must treat the buffers of this solve as in use until it has synchronised the
    // device itself (or given up on it).
    if (rc != MPC_OK && !h->timed_out) {
        const std::string keep = g_err;
        (void)hipDeviceSynchronize();
        g_err = keep;
    }
    return rc;
}
static int run_solver_rounds(mpc_handle *h, hipStream_t s)
{
    const DevCfg &c = h->dc;
    Workspace &w = h->ws;
    const int B = w.B;
    HIPCHK(hipMemsetAsync(w.counts, 0, 8 * MPC_MAX_GROUPS * sizeof(int) + 16 * sizeof(unsigned long long) +
                                           2 * MPC_MAX_GROUPS * sizeof(int), s));
    hipLaunchKernelGGL(init_kernel, dim3((unsigned)(((size_t)B * REC + 255) / 256)), dim3(256), 0, s, c, w);
    if (w.arrive) HIPCHK(hipMemsetAsync(h->arrive_buf, 0, sizeof(int) * (size_t)(w.St / 64 + 16), s)); // (a failed launch may have left counts)
    h->rounds = 0; h->evals_grad = 0; h->evals_cost = 0; h->eval_ms = 0.0; h->step_ms = 0.0;
    h->lbfgs_ms = 0.0; h->lbfgs_rows = 0; h->solo_agents = 0;
    for (int k = 0; k < 5; k++) { h->kernel_ms[k] = 0.0; h->kernel_launches[k] = 0; }
    // Guard against a runaway loop only: a valid solve must never reach it.  An inner iteration costs at
    // most ~(4 + 11 * 60) evaluations (nine line-search trials whose quadratic-upper-bound loop doubles L
    // up to L_max), an outer iteration a handful more; with an evaluation budget an agent stops at the
    // first stop test past it.  Every running agent consumes at least one evaluation per round.
    const long long per_iter = 700;
    long long max_rounds = per_iter * ((long long)c.max_total_inner + 16) + 8LL * c.max_outer + 1024;
    if (c.max_total_evals > 0) max_rounds = std::min(max_rounds, (long long)c.max_total_evals + per_iter + 8LL * c.max_outer + 1024);
    if (h->round_limit > 0) max_rounds = std::min(max_rounds, h->round_limit);   // mpc_set_round_limit (test aid)
    const bool solo_ok = (h->solo_max > 0 || h->solo_all > 0) && solo_fits(h);
    size_t nev = 0;               // events 0 .. nev-1 of the pool: five per sampled launch set
    bool solo_timed[MPC_MAX_GROUPS] = {false};
    auto solo_events = [&](int g, hipStream_t st, int which) { // profile mode: (start, stop) around the launch
        if (!h->profile) return;
        if (!h->soloev[g][which] && hipEventCreate(&h->soloev[g][which]) != hipSuccess) { h->soloev[g][which] = nullptr; return; }
        (void)hipEventRecord(h->soloev[g][which], st);
        if (which == 1 && h->soloev[g][0]) solo_timed[g] = true;
    };
    long long launch_sets = 0, unfused_sets = 0, solo_launches = 0;
    long long rounds_done[MPC_MAX_GROUPS] = {0};
    int ng = 0;
    bool all_solo = false;
    if (solo_ok && B <= h->solo_all) {
        // small batch: every agent is solved by one wave of the persistent kernel from the start
        all_solo = true;
    }
    Workspace gv[MPC_MAX_GROUPS];
    hipStream_t gs[MPC_MAX_GROUPS];
    // groups: contiguous agent ranges (multiples of 64), each with its own stream; measured at
    // B = 65536 (round 1): 1 group 0.258 s, 2 groups 0.224 s, 3 groups 0.220 s per solve.  The HIP runtime
    // maps a process's streams to GPU_MAX_HW_QUEUES hardware queues (4 unless the environment says
    // otherwise): with the caller's stream that leaves three for groups -- a fourth group shares a queue with
    // another and its launches wait behind that one's (round 2: 3 groups 169.8 ms, 4 groups 259.8 ms with 4
    // queues, 165.7 ms with 8; 5 groups 196 ms).  Four groups only when the queues are there.
    int G = h->ngroups > 0 ? h->ngroups : (B >= 49152 && h->hw_queues >= 5 ? 4 : B >= 24576 ? 3 : B >= 16384 ? 2 : 1);
    if (G > MPC_MAX_GROUPS) G = MPC_MAX_GROUPS;
    while (G > 1 && B / G < 1024) G--;
    if (all_solo) G = 1;
    const int per = (((B + G - 1) / G) + 63) & ~63;
    for (int g = 0; g < G; g++) {
        const int lo = g * per, hi = std::min(B, lo + per);
        if (lo >= hi) break;
        gv[ng] = group_view(w, c, ng, lo, hi);
        ng++;
    }
    h->groups_last = ng;
    if (ng == 1) gs[0] = s;
    else {
        if (!h->gevent[MPC_MAX_GROUPS]) HIPCHK(hipEventCreateWithFlags(&h->gevent[MPC_MAX_GROUPS], hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->gevent[MPC_MAX_GROUPS], s)); // fork
        for (int g = 0; g < ng; g++) {
            if (!h->gstream[g]) HIPCHK(hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking));
            if (!h->gevent[g]) HIPCHK(hipEventCreateWithFlags(&h->gevent[g], hipEventDisableTiming));
            gs[g] = h->gstream[g];
            HIPCHK(hipStreamWaitEvent(gs[g], h->gevent[MPC_MAX_GROUPS], 0));
        }
    }
    static const int check_env = getenv("MPC_CHECK_EVERY") ? atoi(getenv("MPC_CHECK_EVERY")) : 0;
    const int check_every = check_env > 0 ? check_env : 8;
    // Every group advances on its own: a window of `check_every` rounds is queued, its request counters are
    // copied back behind it, and the host looks at them ONE WINDOW LATE -- a second window is already queued
    // by then, so the stream does not run dry while the host decides.  The host serves whichever group's
    // counters have arrived (event query, no blocking wait on one group while another's stream empties:
    // the lock-step loop of round 1 left 60 - 200 us bubbles per window on the groups it was not waiting
    // for, ~6 % of their streams' time in the r02d trace).
    struct GroupRun { long long round = 0, window = 0; bool active = true; int slot_bound = 0; };
    GroupRun gr[MPC_MAX_GROUPS];
    // upper bound on a group's requests per round: at most two per running agent (evaluation +
    // speculative gradient); every running agent has at least one request in a round and agents only
    // ever finish, so twice the requests seen at a poll bounds every later round
    for (int g = 0; g < ng; g++) gr[g].slot_bound = 2 * gv[g].B;
    int nactive = ng;
    if (all_solo) {
        solo_events(0, gs[0], 0);
        launch_solo(h, gv[0], gs[0], w.solo_ctr, false, B, max_rounds);
        solo_events(0, gs[0], 1);
        solo_launches++;
        gr[0].active = false; nactive = 0;
    }
    int rc_loop = MPC_OK;
    double host_queue_s = 0.0;                  // host time spent queueing launches (MPC_HOST_TIMING: printed at the end)
    long long dry_windows = 0, first_dry_round = -1;
    static const char *host_trace = getenv("MPC_HOST_TRACE");   // file: one line per polled window (group, round, us, requests)
    std::vector<std::array<long long, 4>> trace;
    static const bool host_timing = getenv("MPC_HOST_TIMING") != nullptr;
    const auto t_loop0 = std::chrono::steady_clock::now();
    auto queue_window_impl = [&](int g) {       // `check_every` rounds of group g, then the copy of its counters
        GroupRun &r = gr[g];
        const Workspace &v = gv[g];
        int cur = 0;
        for (int i = 0; i < check_every && r.round < max_rounds; i++) {
            cur = (int)(r.round & 1);
            int *lists = v.lists + (size_t)cur * 2 * v.Ls;
            int *counts = v.counts + cur * 4;
            int *counts_next = v.counts + (cur ^ 1) * 4;
            hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
            // profile mode samples every 8th round: five events per sampled launch set
            if (h->profile && (r.round & 7) == 0 && get_event(h, nev + 4)) { // all five exist, or none is used
                for (int k = 0; k < 5; k++) ev[k] = h->ev_pool[nev + k];
                nev += 5;
            }
            if (ev[0]) (void)hipEventRecord(ev[0], gs[g]);
            launch_step(h, v, gs[g], lists, counts, counts_next, r.slot_bound, cur);
            if (ev[1]) (void)hipEventRecord(ev[1], gs[g]);
            // (counts[2] of the round's buffer: K1c leaves the number of gradient slots there for the next step
            // kernel's thread-per-agent blocks)
            const bool fused = launch_eval(h, v, gs[g], lists, counts, 0, 0, ev[2], ev[3], r.slot_bound, counts + 2);
            if (ev[4]) (void)hipEventRecord(ev[4], gs[g]);
            r.round++;
            rounds_done[g]++;
            launch_sets++;
            unfused_sets += !fused;
        }
        // (an event query that says "not ready" is recorded as the thread's last error too: not a failure)
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess && le != hipErrorNotReady) { rc_loop = MPC_E_HIP; return; }
        const int wb = (int)(r.window & 1);
        if (!h->pollev[wb][g] && hipEventCreateWithFlags(&h->pollev[wb][g], hipEventDisableTiming) != hipSuccess) { rc_loop = MPC_E_HIP; return; }
        if (hipMemcpyAsync(h->host_counts + 16 * wb + 2 * g, v.counts + cur * 4, 2 * sizeof(int), hipMemcpyDeviceToHost, gs[g]) != hipSuccess ||
            hipEventRecord(h->pollev[wb][g], gs[g]) != hipSuccess) { rc_loop = MPC_E_HIP; return; }
        r.window++;
    };
    auto queue_window = [&](int g) {
        if (!host_timing) { queue_window_impl(g); return; }
        const auto t0 = std::chrono::steady_clock::now();
        queue_window_impl(g);
        host_queue_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    // what the counters of window `pb` say about group g; returns false when the group is done with rounds
    auto decide = [&](int g, int pb) {
        GroupRun &r = gr[g];
        const int reqs = h->host_counts[16 * pb + 2 * g] + h->host_counts[16 * pb + 2 * g + 1];
        if (reqs == 0) return false;
        r.slot_bound = std::min(r.slot_bound, 2 * reqs);
        if (solo_ok && h->solo_max > 0 && reqs <= h->solo_max) {
            // few agents left in this group: they finish in the persistent kernel, each in
            // its own wave, instead of waiting for four launches per evaluation
            solo_events(g, gs[g], 0);
            launch_solo(h, gv[g], gs[g], w.solo_ctr + 2 * g, true, reqs, max_rounds);
            solo_events(g, gs[g], 1);
            solo_launches++;
            return false;
        }
        return true;
    };
    // (tried: the groups started 40 / 80 / 160 us apart, so that one's step kernel meets another's K1 -- no change)
    for (int g = 0; g < ng && nactive > 0; g++) queue_window(g);
    for (int g = 0; g < ng && nactive > 0; g++) if (gr[g].round < max_rounds) queue_window(g);
    // The host has nothing to do while the windows it has queued run (milliseconds with all agents active):
    // it spins on the event queries only for a short while after the last progress, then sleeps in short naps
    // -- a second window is always queued behind the one polled, so a nap delays no launch -- and leaves its
    // core to whoever needs it (eight ranks on one node are eight of these loops: INTEGRATION.md 4).
    // MPC_SPIN=1 keeps the pure busy-wait.
    static const bool spin_only = getenv("MPC_SPIN") != nullptr;
    auto last_progress = std::chrono::steady_clock::now();
    while (nactive > 0 && rc_loop == MPC_OK) {
        bool progressed = false;
        for (int g = 0; g < ng; g++) {
            GroupRun &r = gr[g];
            if (!r.active) continue;
            const long long oldest = r.window - (r.window >= 2 ? 2 : 1);    // the window whose counters are looked at next
            const int pb = (int)(oldest & 1);
            const hipError_t q = hipEventQuery(h->pollev[pb][g]);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) { rc_loop = MPC_E_HIP; break; }
            progressed = true;
            if (host_timing && r.window - oldest > 1 && hipEventQuery(h->pollev[pb ^ 1][g]) == hipSuccess) {
                // both queued windows have run: this group's stream was empty while the host was elsewhere
                if (dry_windows++ == 0) first_dry_round = r.round;
            }
            if (host_trace)
                trace.push_back({(long long)g, r.round - (r.window - oldest) * check_every + check_every - 1,
                                 (long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_loop0).count(),
                                 (long long)(h->host_counts[16 * pb + 2 * g] + h->host_counts[16 * pb + 2 * g + 1])});
            bool go = decide(g, pb);
            if (go && r.round >= max_rounds) {
                // the round limit: nothing more can be queued; the verdict is the LAST window's
                if (r.window - oldest > 1) {
                    if (hipEventSynchronize(h->pollev[pb ^ 1][g]) != hipSuccess) { rc_loop = MPC_E_HIP; break; }
                    go = decide(g, pb ^ 1);
                }
                if (go) { rc_loop = MPC_E_LIMIT; break; }
            }
            if (!go) { r.active = false; nactive--; continue; }
            queue_window(g);
        }
        if (progressed) { last_progress = std::chrono::steady_clock::now(); continue; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - last_progress).count() > h->poll_timeout_s) {
            h->timed_out = true; rc_loop = MPC_E_HIP; break;      // no window has completed for poll_timeout_s
        }
        if (spin_only || std::chrono::steady_clock::now() - last_progress < std::chrono::microseconds(40))
            __builtin_ia32_pause();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    if (host_timing)
        fprintf(stderr, "[mpc host] round loop %.2f ms, of which queueing launches %.2f ms (%lld launch sets, %d groups); "
                        "windows found with the stream already empty: %lld (first at round %lld)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count(),
                host_queue_s * 1e3, launch_sets, ng, dry_windows, first_dry_round);
    if (host_trace) {
        if (FILE *f = fopen(host_trace, "a")) {
            fprintf(f, "# solve: group, last round of the window, us since the loop began, requests of that round\n");
            for (const auto &t : trace) fprintf(f, "%lld %lld %lld %lld\n", t[0], t[1], t[2], t[3]);
            fclose(f);
        }
    }
    if (rc_loop == MPC_E_LIMIT) return fail(MPC_E_LIMIT, "mpc_solve_batch: round limit reached");
    if (h->timed_out)
        return fail(MPC_E_HIP, "mpc_solve_batch: wall-clock bound of " + std::to_string(h->poll_timeout_s) +
                               " s expired in the round loop: no polled window completed (mpc_set_poll_timeout); work is still queued");
    if (rc_loop != MPC_OK) return fail(rc_loop, "mpc_solve_batch: HIP error in the round loop");
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess && le != hipErrorNotReady) return fail(MPC_E_HIP, std::string("mpc_solve_batch: ") + hipGetErrorString(le));
    }
    if (ng > 1) { // join
        for (int g = 0; g < ng; g++) {
            HIPCHK(hipEventRecord(h->gevent[g], gs[g]));
            HIPCHK(hipStreamWaitEvent(s, h->gevent[g], 0));
        }
    }
    for (int g = 0; g < ng; g++) h->rounds = std::max<int64_t>(h->rounds, rounds_done[g]);
    {
        unsigned long long *tot = (unsigned long long *)((char *)h->host_counts + 128);   // pinned (see host_counts)
        int *sctr = (int *)((char *)h->host_counts + 256);
        static_assert(16 * sizeof(unsigned long long) == 128 && 2 * MPC_MAX_GROUPS * sizeof(int) == 64, "pinned staging layout");
        hipLaunchKernelGGL(totals_kernel, grid_for(B, 256), dim3(256), 0, s, w);
        HIPCHK(hipMemcpyAsync(tot, w.totals, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(sctr, w.solo_ctr, 2 * MPC_MAX_GROUPS * sizeof(int), hipMemcpyDeviceToHost, s));
        { const int rs = bounded_sync(h, s, "mpc_solve_batch"); if (rs) return rs; }
        h->evals_grad = (int64_t)tot[0]; h->evals_cost = (int64_t)tot[1]; h->lbfgs_rows = (int64_t)tot[2];
        h->spec_issued = (int64_t)tot[4]; h->spec_used = (int64_t)tot[5];
        h->la_evals = (int64_t)tot[7]; h->la_hits = (int64_t)tot[8];
        if (all_solo) h->solo_agents = B;
        else for (int g = 0; g < ng; g++) h->solo_agents += sctr[2 * g + 1];
        // every agent must have reached PH_DONE: the round path says so through its request counters, the
        // persistent kernel only through the records (its trip guard leaves an agent where it stands)
        if (tot[6] != 0)
            return fail(MPC_E_LIMIT, "mpc_solve_batch: round limit reached (" + std::to_string(tot[6]) +
                                     " agents unfinished in the persistent kernel)");
    }
    if (h->profile) {
        { const int rs = bounded_sync(h, s, "mpc_solve_batch"); if (rs) return rs; }
        for (size_t i = 0; i + 4 < nev; i += 5) {
            for (int k = 0; k < 4; k++) {
                float d = 0.f;
                (void)hipEventElapsedTime(&d, h->ev_pool[i + k], h->ev_pool[i + k + 1]);
                h->kernel_ms[k] += d;
            }
        }
        // scale the sampled sums to all launch sets of the solve
        const double sampled = (double)(nev / 5);
        const double scale = sampled > 0 ? (double)launch_sets / sampled : 0.0;
        for (int k = 0; k < 4; k++) h->kernel_ms[k] *= scale;
        h->solo_longest_ms = 0.0;
        for (int g = 0; g < MPC_MAX_GROUPS; g++) {
            float d = 0.f;
            if (solo_timed[g]) (void)hipEventElapsedTime(&d, h->soloev[g][0], h->soloev[g][1]);
            h->kernel_ms[4] += d;                       // summed over the groups (their launches overlap in time)
            h->solo_longest_ms = std::max(h->solo_longest_ms, (double)d);
        }
        h->step_ms = h->kernel_ms[0];
        h->eval_ms = h->kernel_ms[1] + h->kernel_ms[2] + h->kernel_ms[3];
    }
    h->launches = (int64_t)launch_sets;
    h->kernel_launches[0] = h->kernel_launches[1] = h->kernel_launches[2] = launch_sets;
    h->kernel_launches[3] = unfused_sets;
    h->kernel_launches[4] = solo_launches;
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

static int solve_batch_impl(mpc_handle *h, int B, const double *x0, const double *cl, const int32_t *cl_index,
                            double *U, double *lambda, double *stats, void *stream, bool from_worker);
extern "C" int mpc_solve_batch(mpc_handle *h, int B, const double *x0, const double *cl,
                               const int32_t *cl_index, double *U, double *lambda, double *stats,
                               void *stream)
{
    return solve_batch_impl(h, B, x0, cl, cl_index, U, lambda, stats, stream, false);
}
static int solve_batch_impl(mpc_handle *h, int B, const double *x0, const double *cl, const int32_t *cl_index,
                            double *U, double *lambda, double *stats, void *stream, bool from_worker)
{
    int rc = check_common(h, B, "mpc_solve_batch", from_worker); if (rc) return rc;
    if (B == 0) return MPC_OK;
    if (!x0 || !cl || !U) return fail(MPC_E_ARG, "mpc_solve_batch: null buffer");
    const DevCfg &c = h->dc;
    if (c.m && !lambda) return fail(MPC_E_ARG, "mpc_solve_batch: lambda is required when m > 0");
    rc = reserve(h, B); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Workspace &w = h->ws;
    w.cl = cl; w.cl_index = cl_index; w.x0 = x0; w.xo = U; w.y = lambda; w.psi_direct = nullptr;
    w.near = near_for(h, cl);
    w.arrive = h->arrive_adjoint ? h->arrive_buf : nullptr;
    w.xe = w.ws_xe; w.ge = w.ws_ge; w.yhe = w.ws_yhe; w.Sig = w.ws_Sig;
    rc = run_solver(h, s); if (rc) return rc;
    if (stats) hipLaunchKernelGGL(stats_kernel, grid_for(B, 256), dim3(256), 0, s, w, stats);
    HIPCHK(hipGetLastError());
    return bounded_sync(h, s, "mpc_solve_batch");
}

// The round loop of a solve is host code (launches, counter polls): the asynchronous form runs it on a
// worker thread owned by the handle, so the caller gets its thread back at once and may run a second
// handle's solve, or its own work, beside it.  One solve in flight per handle.
static void async_worker(mpc_handle *h)
{
    for (;;) {
        mpc_handle::AsyncJob j;
        {
            std::unique_lock<std::mutex> lk(h->mu);
            h->cv.wait(lk, [&] { return h->job_posted || h->worker_quit; });
            if (!h->job_posted) return;   // quit
            j = h->job; h->job_posted = false; h->job_running = true;
        }
        const int rc = solve_batch_impl(h, j.B, j.x0, j.cl, j.cl_index, j.U, j.lambda, j.stats, j.stream, true);
        {
            std::lock_guard<std::mutex> lk(h->mu);
            h->job_rc = rc; h->job_err = rc ? g_err : std::string();
            h->job_running = false; h->job_done = true;
        }
        h->cv.notify_all();
    }
}

extern "C" int mpc_solve_batch_async(mpc_handle *h, int B, const double *x0, const double *cl,
                                     const int32_t *cl_index, double *U, double *lambda, double *stats,
                                     void *stream)
{
    if (!h) return fail(MPC_E_ARG, "mpc_solve_batch_async: null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->job_posted || h->job_running || h->job_done)
        return fail(MPC_E_ARG, "mpc_solve_batch_async: a solve of this handle is in flight (mpc_solve_wait first)");
    if (!h->worker.joinable()) h->worker = std::thread(async_worker, h);
    h->job = {B, x0, cl, cl_index, U, lambda, stats, stream};
    h->job_posted = true;
    h->cv.notify_all();
    return MPC_OK;
}

extern "C" int mpc_solve_wait(mpc_handle *h)
{
    if (!h) return fail(MPC_E_ARG, "mpc_solve_wait: null handle");
    std::unique_lock<std::mutex> lk(h->mu);
    if (!h->job_posted && !h->job_running && !h->job_done) return fail(MPC_E_ARG, "mpc_solve_wait: no solve in flight");
    h->cv.wait(lk, [&] { return h->job_done; });
    h->job_done = false;
    if (h->job_rc != MPC_OK) g_err = h->job_err;
    return h->job_rc;
}

extern "C" int mpc_closed_loop(mpc_handle *h, int B, int T, int shift, double *x, const double *cl,
                               const int32_t *cl_index, double *U, double *lambda, double *traj_x,
                               double *traj_u, int32_t *fail_count, double *stats, void *stream)
{
    int rc = check_common(h, B, "mpc_closed_loop"); if (rc) return rc;
    if (B == 0 || T == 0) return MPC_OK;
    if (T < 0 || !x || !cl || !U) return fail(MPC_E_ARG, "mpc_closed_loop: bad argument");
    const DevCfg &c = h->dc;
    if (c.m && !lambda) return fail(MPC_E_ARG, "mpc_closed_loop: lambda is required when m > 0");
    hipStream_t s = (hipStream_t)stream;
    double *st = stats;
    if (!st) {
        rc = reserve_stage(h, sizeof(double) * 8 * (size_t)B); if (rc) return rc;
        st = h->stage;
    }
    for (int t = 0; t < T; t++) {
        rc = mpc_solve_batch(h, B, x, cl, cl_index, U, lambda, st, stream); if (rc) return rc;
        if (c.model == PAC)
            hipLaunchKernelGGL(plant_step_kernel<PAC>, grid_for(B, 64), dim3(64), 0, s, c, B, t, T, shift, x, U,
                               traj_x, traj_u, st, fail_count);
        else
            hipLaunchKernelGGL(plant_step_kernel<KIN>, grid_for(B, 64), dim3(64), 0, s, c, B, t, T, shift, x, U,
                               traj_x, traj_u, st, fail_count);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return MPC_OK;
}

extern "C" int mpc_last_speculation(mpc_handle *h, int64_t *issued, int64_t *used)
{
    if (!h) return fail(MPC_E_ARG, "mpc_last_speculation: null handle");
    { const int rb = refuse_if_busy(h, "mpc_last_speculation"); if (rb) return rb; }
    if (issued) *issued = h->spec_issued;
    if (used) *used = h->spec_used;
    return MPC_OK;
}

extern "C" int mpc_last_lookahead(mpc_handle *h, int64_t *evals, int64_t *hits)
{
    if (!h) return fail(MPC_E_ARG, "mpc_last_lookahead: null handle");
    { const int rb = refuse_if_busy(h, "mpc_last_lookahead"); if (rb) return rb; }
    if (evals) *evals = h->la_evals;
    if (hits) *hits = h->la_hits;
    return MPC_OK;
}

extern "C" int mpc_last_solve_info2(mpc_handle *h, double *launch_pairs, int64_t *lbfgs_rows)
{
    if (!h) return fail(MPC_E_ARG, "mpc_last_solve_info2: null handle");
    { const int rb = refuse_if_busy(h, "mpc_last_solve_info2"); if (rb) return rb; }
    if (launch_pairs) *launch_pairs = (double)h->launches; // (step, eval) launch sets of the last solve
    if (lbfgs_rows) *lbfgs_rows = h->lbfgs_rows;
    return MPC_OK;
}

extern "C" int mpc_last_solve_info(mpc_handle *h, int64_t *rounds, int64_t *evals_grad, int64_t *evals_cost,
                                   double *eval_ms, double *step_ms)
{
    if (!h) return fail(MPC_E_ARG, "mpc_last_solve_info: null handle");
    { const int rb = refuse_if_busy(h, "mpc_last_solve_info"); if (rb) return rb; }
    if (rounds) *rounds = h->rounds;
    if (evals_grad) *evals_grad = h->evals_grad;
    if (evals_cost) *evals_cost = h->evals_cost;
    if (eval_ms) *eval_ms = h->eval_ms;
    if (step_ms) *step_ms = h->step_ms;
    return MPC_OK;
}

extern "C" int mpc_last_kernel_ms(mpc_handle *h, double *out4)
{
    if (!h || !out4) return fail(MPC_E_ARG, "mpc_last_kernel_ms: null argument");
    { const int rb = refuse_if_busy(h, "mpc_last_kernel_ms"); if (rb) return rb; }
    for (int k = 0; k < 4; k++) out4[k] = h->kernel_ms[k];
    return MPC_OK;
}

extern "C" int mpc_last_kernel_profile(mpc_handle *h, double *ms5, int64_t *launches5, int64_t *solo_agents)
{
    if (!h) return fail(MPC_E_ARG, "mpc_last_kernel_profile: null handle");
    { const int rb = refuse_if_busy(h, "mpc_last_kernel_profile"); if (rb) return rb; }
    for (int k = 0; k < 5; k++) {
        if (ms5) ms5[k] = h->kernel_ms[k];
        if (launches5) launches5[k] = h->kernel_launches[k];
    }
    if (solo_agents) *solo_agents = h->solo_agents;
    return MPC_OK;
}

extern "C" int mpc_set_nearest_blocks(mpc_handle *h, int on)
{
    if (!h) return fail(MPC_E_ARG, "mpc_set_nearest_blocks: null handle");
    { const int rb = refuse_if_busy(h, "mpc_set_nearest_blocks"); if (rb) return rb; }
    if (on < 0 || on > 2) return fail(MPC_E_ARG, "mpc_set_nearest_blocks: mode is 0 (full scan), 1 (block boxes) or 2 (grid)");
    h->nearest_mode = on;
    return MPC_OK;
}

extern "C" int mpc_set_solo_max(mpc_handle *h, int max_requests)
{
    if (!h || max_requests < 0) return fail(MPC_E_ARG, "mpc_set_solo_max: bad argument");
    { const int rb = refuse_if_busy(h, "mpc_set_solo_max"); if (rb) return rb; }
    h->solo_max = h->solo_all = max_requests;
    return MPC_OK;
}

extern "C" int mpc_last_solo_ms(mpc_handle *h, double *sum_ms, double *longest_ms)
{
    if (!h) return fail(MPC_E_ARG, "mpc_last_solo_ms: null handle");
    { const int rb = refuse_if_busy(h, "mpc_last_solo_ms"); if (rb) return rb; }
    if (sum_ms) *sum_ms = h->kernel_ms[4];
    if (longest_ms) *longest_ms = h->solo_longest_ms;
    return MPC_OK;
}

extern "C" int mpc_stream_concurrency(mpc_handle *h, int *streams, int *groups_last)
{
    if (!h) return fail(MPC_E_ARG, "mpc_stream_concurrency: null handle");
    { const int rb = refuse_if_busy(h, "mpc_stream_concurrency"); if (rb) return rb; }
    if (streams) *streams = h->hw_queues;
    if (groups_last) *groups_last = h->groups_last;
    return MPC_OK;
}

extern "C" int mpc_set_round_limit(mpc_handle *h, int64_t rounds)
{
    if (!h || rounds < 0) return fail(MPC_E_ARG, "mpc_set_round_limit: bad argument");
    { const int rb = refuse_if_busy(h, "mpc_set_round_limit"); if (rb) return rb; }
    h->round_limit = (long long)rounds;
    return MPC_OK;
}

extern "C" int mpc_set_memo(mpc_handle *h, int on)
{
    if (!h) return fail(MPC_E_ARG, "mpc_set_memo: null handle");
    { const int rb = refuse_if_busy(h, "mpc_set_memo"); if (rb) return rb; }
    h->dc.no_memo = on ? 0 : 1;
    return MPC_OK;
}

extern "C" int mpc_set_groups(mpc_handle *h, int groups)
{
    if (!h || groups < 0 || groups > MPC_MAX_GROUPS) return fail(MPC_E_ARG, "mpc_set_groups: bad argument");
    { const int rb = refuse_if_busy(h, "mpc_set_groups"); if (rb) return rb; }
    h->ngroups = groups;
    return MPC_OK;
}

extern "C" int mpc_set_profile(mpc_handle *h, int on)
{
    if (!h) return fail(MPC_E_ARG, "mpc_set_profile: null handle");
    { const int rb = refuse_if_busy(h, "mpc_set_profile"); if (rb) return rb; }
    h->profile = on != 0;
    return MPC_OK;
}

extern "C" int mpc_set_poll_timeout(mpc_handle *h, double seconds)
{
    if (!h || !(seconds > 0.0)) return fail(MPC_E_ARG, "mpc_set_poll_timeout: bad argument");
    { const int rb = refuse_if_busy(h, "mpc_set_poll_timeout"); if (rb) return rb; }
    h->poll_timeout_s = seconds;
    return MPC_OK;
}

// test aid: the library's idling kernel (one wave that sleeps until `microseconds` of the device's wall clock have
// passed) queued on `stream` -- work that holds a stream for a known time (the wall-clock bound's test)
extern "C" int mpc_debug_spin(mpc_handle *h, double microseconds, void *stream)
{
    int rc = check_common(h, 0, "mpc_debug_spin"); if (rc) return rc;
    if (!(microseconds >= 0.0) || microseconds > 30e6) return fail(MPC_E_ARG, "mpc_debug_spin: 0 .. 30 s");
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) khz = 100000;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(microseconds * 1e-6 * khz * 1e3));
    HIPCHK(hipGetLastError());
    return MPC_OK;
}

// diagnostic: the per-agent solver records as the last solve left them (after a solve stopped by max_total_inner = k
// they hold the state at the top of inner iteration k: step sizes, line-search step, active-set size, history
// fill, counters), decoded to plain doubles, into a HOST array [B][MPC_NREC]; names: mpc_debug_record_names()
static const char *const k_record_names =
    "psi,L,gamma,phi,psixh,pp,gp,tau,psin,Ln,gamman,psixhn,gpn,ppn,sigpp,eps,hn2,hfd,gamma_top,Delta,rho,eps_old,ne1,ps_eps,"
    "out_eps,out_delta,psi_out,psie,phase,k,lidx,lfull,noprog,nJ,outer,first,initred,penred,inner_tot,inner_fail,status,"
    "nevals,maxit,overwrite,fallback,ps_status,ps_iters,out_of_iter,ngrad,lbrows,spec,spec_gamma,nspec,nspec_used,ncost,"
    "run_mineps,run_ev0,memo_status,memo_iters,memo_evals,memo_mineps,memo_eps,la_evals,la_hits";
static_assert(R_USED == 64 && REC == MPC_NREC, "k_record_names / MPC_NREC must follow the record enum");
extern "C" const char *mpc_debug_record_names(void) { return k_record_names; }
extern "C" int mpc_debug_records(mpc_handle *h, int B, double *host_out)
{
    int rc = check_common(h, B, "mpc_debug_records"); if (rc) return rc;
    if (!host_out) return fail(MPC_E_ARG, "mpc_debug_records: null buffer");
    if (B == 0) return MPC_OK;
    if (!h->arena || B > h->ws.B) return fail(MPC_E_ARG, "mpc_debug_records: more agents than the last solve held");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host_out, h->ws.rec, sizeof(double) * (size_t)B * REC, hipMemcpyDeviceToHost));
    for (int a = 0; a < B; a++)
        for (int sl = 0; sl < REC; sl++)
            if (rec_is_int(sl)) {
                int64_t bits; std::memcpy(&bits, &host_out[(size_t)a * REC + sl], 8);
                host_out[(size_t)a * REC + sl] = (double)(int32_t)(bits & 0xffffffffLL);
            }
    return MPC_OK;
}

#ifdef MPC_DEV_STAMP
extern "C" int mpc_dev_records(mpc_handle *h, double *host_out)   // (experiment) the per-agent records as the last solve left them
{
    (void)hipDeviceSynchronize();
    return hipMemcpy(host_out, h->ws.rec, sizeof(double) * (size_t)h->ws.B * mpc::REC, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
extern "C" int mpc_dev_stamps(long long *host_out)       // (timing experiment) the stamps of the last launch of the kernel chosen
{
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mpc::g_dev_stamps), sizeof(long long) * 4 * mpc::DEV_STAMPS) == hipSuccess ? 0 : 1;
}
#endif
