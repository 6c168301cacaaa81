/*
 * mpc_hip.h -- C-ABI of libmpc_hip.so, the MI355X (gfx950) batched MPC solve step.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (pure Python) reaches its solver through pybind11/dlopen:
 *     controller.py:27-49   pa.ALMSolver(alm_params, inner_solver=pa.StructuredPANOCLBFGSSolver(..))
 *     controller.py:57      x, y, stats = solver(problem, x, y)
 *     main.py:54-56         generate_and_compile_casadi_problem(f, g); prob.C.lowerbound/upperbound
 *     car_dynamics.py:159   f_d.mapaccum(N)(y0, u, p)          (rollout / plant step)
 * A maintainer binds the entry points below with ctypes.
 *
 * Conventions
 *  - plain pointers and sizes only; every data pointer is a DEVICE pointer (HBM) owned by
 *    the caller; the library owns only its handle and its scratch workspace.
 *  - batch arrays are agent-major row-major: x0 [B][nx], U [B][2N] with the reference's
 *    flat stage-major order [d0, delta0, d1, delta1, ...] (car_dynamics.py:149-157),
 *    lambda [B][m], stats [B][MPC_NSTATS].
 *  - centerlines: table cl [C][2S], each row flat [x_0..x_{S-1}, y_0..y_{S-1}]
 *    (main.py:113 ravel(order='F')); cl_index [B] int32 selects a row per agent, NULL = row 0.
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*) except
 *    mpc_solve_batch / mpc_closed_loop, which poll device counters and return when the batch is solved.
 *  - between mpc_solve_batch_async and mpc_solve_wait the handle belongs to its worker thread: every
 *    other call on it returns MPC_E_ARG without touching it.
 *  - return value: 0 on success, negative MPC_E_* otherwise; mpc_last_error() explains.
 *    No exceptions cross the ABI.  One handle per GPU/stream; a handle is not thread-safe.
 */
#ifndef MPC_HIP_H
#define MPC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MPC_MODEL_KINEMATIC = 0, /* dynamics.py:122-173, nx = 4 [x, y, phi, v] */
       MPC_MODEL_PACEJKA = 1 }; /* car_dynamics.py:93-129, nx = 6 [x, y, phi, vx, vy, omega] */
enum { MPC_WRAP_FLOOR = 0, MPC_WRAP_FMOD = 1, MPC_WRAP_IEEE = 2 }; /* car_dynamics.py:168-172 */
enum { MPC_CONSTR_NONE = 0,     /* m = 0 (== main.py:57 left commented: D = R^m) */
       MPC_CONSTR_STATE_SQ = 1, /* main.py:43-52: g = x_i^2 - off_i per stage, bounds D_lb/D_ub */
       MPC_CONSTR_LANE = 2 };   /* signed lateral distance to the centerline within +-halfwidth */
/* alpaqa SolverStatus values read at controller.py:59-64 */
enum { MPC_ST_UNKNOWN = 0, MPC_ST_CONVERGED = 1, MPC_ST_MAXTIME = 2, MPC_ST_MAXITER = 3,
       MPC_ST_NOTFINITE = 4, MPC_ST_NOPROGRESS = 5, MPC_ST_INTERRUPTED = 6 };
enum { MPC_OK = 0, MPC_E_ARG = -1, MPC_E_HIP = -2, MPC_E_ALLOC = -3, MPC_E_LIMIT = -4 };

#define MPC_NSTATS 8 /* [status, outer_iters, inner_iters, inner_failures, eps, delta, psi, n_evals] */
#define MPC_MAX_N 64 /* horizon limit */
#define MPC_NREC 64  /* doubles per agent of mpc_debug_records */

typedef struct mpc_config {
    int32_t model;           /* MPC_MODEL_* */
    int32_t N;               /* horizon (main.py:68) */
    int32_t S;               /* centerline points (main.py:70) */
    int32_t nfe;             /* RK4 finite elements per stage (car_dynamics.py:136) */
    int32_t wrap_mode;       /* MPC_WRAP_* */
    int32_t clip_inputs;     /* dynamics.py:99,:163 np.clip of the inputs inside the RHS */
    int32_t constr_mode;     /* MPC_CONSTR_* */
    int32_t lbfgs_memory;    /* controller.py:36 */
    int32_t max_iter;        /* controller.py:31 */
    int32_t max_outer;       /* controller.py:45 */
    int32_t hess_heuristic;  /* controller.py:32 */
    int32_t max_no_progress; /* alpaqa default 10 */
    double Ts;               /* car_dynamics.py:93 */
    double v_ref;            /* main.py:65 */
    double cost_w[6];        /* car_dynamics.py:230 */
    double veh[22];          /* car_dynamics.py:65-88 order; main.py:82-111 values */
    double accel, friction;  /* dynamics.py:34-35 */
    double u_lb[2], u_ub[2]; /* main.py:55-56, order [d, delta] */
    double g_off[6];         /* main.py:46-51 */
    double D_lb[6], D_ub[6]; /* main.py:57 */
    double lane_halfwidth;
    double alm_eps, alm_delta, Sigma0, eps0, rho, Delta, theta, M, Sigma_max; /* controller.py:40-43 */
    double Delta_lower, Sigma0_lower, eps0_increase, rho_increase;
    int32_t max_num_initial_retries, max_num_retries, max_total_num_retries;
    int32_t max_total_inner; /* iteration budget replacing controller.py:30,:44 wall-clock caps */
    int32_t max_total_evals; /* evaluation budget (0 = none), checked in the inner stop test where
                                alpaqa checks the clock: status MaxTime (2) when hit */
    double lip_eps, lip_delta, Lgamma_factor, L_min, L_max, tau_min, qub_tol;
} mpc_config;

typedef struct mpc_handle mpc_handle;

/* fills *cfg with the reference's constants (main.py:65-111, controller.py:27-48) */
int mpc_default_config(mpc_config *cfg, int model, int N);
int mpc_nx(const mpc_config *cfg);
int mpc_m(const mpc_config *cfg);

/* replaces controller.py:12-49 (solver construction) + main.py:25-59 (problem construction) */
int mpc_create(const mpc_config *cfg, int device, mpc_handle **out);
int mpc_destroy(mpc_handle *h);
const char *mpc_last_error(void);
/* identity of the build: SHA-256 over the library's sources and this header as _lib.build() compiled them
 * ("unknown" for a build by other means).  Committed profiles name the build they measured by it and bench.py
 * uses a profile's counters only when it matches the running library. */
const char *mpc_source_hash(void);

/* a-1 (car_dynamics.py:93-132 / dynamics.py:67-119,:144-173): dx[B][nx] = f(x[B][nx], u[B][2]) */
int mpc_rhs(mpc_handle *h, int B, const double *x, const double *u, double *dx, void *stream);

/* a-2/a-3 (car_dynamics.py:134-147,:159-166 simulate/mapaccum): X[B][Nsim][nx] = x_1..x_Nsim;
 * U is [B][2*Nsim].  Nsim = 1 is the plant step of main.py:145. */
int mpc_rollout(mpc_handle *h, int B, int Nsim, const double *x0, const double *U, double *X,
                void *stream);

/* f-2 (car_dynamics.py:185-190, the 98-candidate scan per stage): prepares the exact pruned
 * nearest-point searches for the centerline table cl [C][2S], on `stream`: (mode 2) a grid of index
 * ranges per row -- every cell holds the range [lo, hi] guaranteed to contain the scan's answer for any
 * point of the cell -- and (mode 1) bounding boxes of blocks of 8 consecutive points.  Later calls that
 * are handed the SAME table pointer (mpc_solve_batch, mpc_eval_cost_grad*, mpc_stage_errors,
 * mpc_closed_loop) use the search mpc_set_nearest_blocks selects; any other table takes the full scan.
 * The index found is the same whichever runs, bit for bit.  Call it again whenever the table's contents
 * change.  mpc_set_nearest_blocks(h, mode): 0 full scan (environment MPC_NEAREST_SCAN), 1 block boxes
 * (MPC_NEAREST_BLOCKS; measured 3 % slower than the scan), 2 grid (default; measured 4.8 % faster
 * solves: DESIGN.md 8). */
int mpc_centerline_blocks(mpc_handle *h, const double *cl, int C, void *stream);
int mpc_set_nearest_blocks(mpc_handle *h, int on);

/* a-4/a-5 (car_dynamics.py:174-228): pose[B][3] = [x, y, phi] -> err[B][3] = [cte, heading_error,
 * pos_error], idx[B] = nearest index (idx may be NULL).  Diagnostics of main.py:122-133. */
int mpc_stage_errors(mpc_handle *h, int B, const double *pose, const double *cl,
                     const int32_t *cl_index, double *err, int32_t *idx, void *stream);

/* a-6 (car_dynamics.py:230-258 L_cost): out[B] = stage cost of (x[B][nx], u[B][2]) */
int mpc_stage_cost(mpc_handle *h, int B, const double *x, const double *u, const double *cl,
                   const int32_t *cl_index, double *out, void *stream);

/* a-6..a-9, kernel K1: psi[B] = f(U) + 1/2 dist_Sigma^2(g(U)+y/Sigma, D); grad[B][2N] (NULL: cost
 * only); yhat[B][m] (NULL ok).  y, Sigma [B][m] are ignored when m == 0.
 * Replaces the CasADi-generated f / grad_f / g / grad_g_prod that alpaqa calls (main.py:54). */
int mpc_eval_cost_grad(mpc_handle *h, int B, const double *x0, const double *cl,
                       const int32_t *cl_index, const double *U, const double *y,
                       const double *Sigma, double *psi, double *grad, double *yhat, void *stream);

/* the same evaluation by the wave-per-agent code of the persistent solve kernel (one wavefront per
 * agent: wide rollout, stage k on lane k, adjoint on one lane) -- bit-identical results */
int mpc_eval_cost_grad_wave(mpc_handle *h, int B, const double *x0, const double *cl,
                            const int32_t *cl_index, const double *U, const double *y,
                            const double *Sigma, double *psi, double *grad, double *yhat, void *stream);

/* a-10, kernel K2: forward-backward step. p = clamp(-gamma*grad, lb-x, ub-x), xhat = x+p;
 * out[B][2] = [||p||^2, grad'p].  gamma[B]. */
int mpc_prox_step(mpc_handle *h, int B, const double *x, const double *grad, const double *gamma,
                  double *xhat, double *p, double *out, void *stream);

/* a-11, kernel K3: masked L-BFGS two-loop.  S,Y [B][M][n] history (row i of agent b is pair i),
 * idx[B] = next write slot, full[B] = ring full flag, mask[B][n] (1 = index in J), q[B][n] inout,
 * ok[B] out (0 when no valid pair: q untouched). */
int mpc_lbfgs_apply(mpc_handle *h, int B, const double *S, const double *Y, const int32_t *idx,
                    const int32_t *full, const double *mask, double *q, int32_t *ok, void *stream);

/* a-8..a-13: the batched solve.  U [B][2N] and lambda [B][m] are warm start in / solution out
 * (controller.py:57); stats [B][MPC_NSTATS].  Replaces `self.solver(self.problem, self.U, self.lam)`. */
int mpc_solve_batch(mpc_handle *h, int B, const double *x0, const double *cl,
                    const int32_t *cl_index, double *U, double *lambda, double *stats,
                    void *stream);

/* The same solve without holding the caller's thread: the round loop (launches and counter polls, host
 * code) runs on a worker thread of the handle; mpc_solve_wait blocks until the solve is complete (all
 * results in the caller's buffers, the stream drained) and returns its code.  One solve in flight per
 * handle; no other call on the handle between the two.  ("async on the given stream".) */
int mpc_solve_batch_async(mpc_handle *h, int B, const double *x0, const double *cl,
                          const int32_t *cl_index, double *U, double *lambda, double *stats,
                          void *stream);
int mpc_solve_wait(mpc_handle *h);

/* f-1 (main.py:121-146): T closed-loop steps with states, controls and warm starts resident on the
 * device: per step one mpc_solve_batch (whose round loop is HOST code: launches and counter polls, like
 * any solve) followed by one kernel that applies u0 and advances the plant with f_d.  Data never leaves
 * the device; control returns to the host once per round window, as in every solve.
 * x [B][nx] inout; U, lambda warm start inout; traj_x [B][T][nx], traj_u [B][T][2] (NULL ok);
 * shift != 0 shifts the warm start by one stage (the reference does not: controller.py:57).
 * fail_count[B] int32 accumulates status != Converged (controller.py:64), NULL ok. */
int mpc_closed_loop(mpc_handle *h, int B, int T, int shift, double *x, const double *cl,
                    const int32_t *cl_index, double *U, double *lambda, double *traj_x,
                    double *traj_u, int32_t *fail_count, double *stats, void *stream);

/* profiling aid: rounds (eval launches) and kernel time of the last mpc_solve_batch */
int mpc_last_solve_info(mpc_handle *h, int64_t *rounds, int64_t *evals_grad, int64_t *evals_cost,
                        double *eval_ms, double *step_ms);
/* f-3 (game_theory.py:115-244 Car.get_total_payoff and its parts): lane-change payoffs of B traffic
 * scenes.  params15 (HOST array) = [L, W, l, theta_max, tlc, td, ti, tau, a_max, h, Lf, q1, q2, a, b]
 * (game_theory.py:23-40,:115,:205); ego [B][3] = (x, v, lane); cars [B][K][3]; ncars [B] <= K <= 62;
 * out [B][2][4] = target lane 1, 2 -> [total, safety, velocity, comfort]. */
int mpc_lane_payoff(mpc_handle *h, int B, int K, const double *params15, const double *ego,
                    const double *cars, const int32_t *ncars, double *out, void *stream);

/* test aid: evaluates the device math used by the kernels; op 0 sin, 1 cos, 2 atan, 3 atan2(a,b),
 * 4 tan on n values */
int mpc_math_probe(mpc_handle *h, int n, int op, const double *a, const double *b, double *out,
                   void *stream);
/* more figures of the last solve: launch pairs (step, eval) issued over all sub-batch groups and
 * L-BFGS history pairs read by K3 */
int mpc_last_solve_info2(mpc_handle *h, double *launch_pairs, int64_t *lbfgs_rows);
/* speculation of the last solve: gradients evaluated ahead of need on the second channel (the
 * next iteration's Hessian-vector point, assuming the line-search trial is accepted) and how many of
 * them the next iteration consumed */
int mpc_last_speculation(mpc_handle *h, int64_t *issued, int64_t *used);
/* lookahead of the persistent kernel (Pacejka model, N <= 16, no constraints; environment MPC_NO_LOOKAHEAD at
 * mpc_create turns it off; results do not depend on it): evaluations of points the state machine was GOING to ask
 * for (next line-search trial points, deeper descent-lemma levels) executed in idle lanes beside a requested
 * evaluation, and requests later served from them without an evaluation trip */
int mpc_last_lookahead(mpc_handle *h, int64_t *evals, int64_t *hits);
/* profile mode: summed HIP-event durations (ms) of the last solve's kernels,
 * out4 = [step_kernel, rollout_kernel (K1a), stage_kernel (K1b), adjoint_kernel (K1c)] */
int mpc_last_kernel_ms(mpc_handle *h, double *out4);
/* the same with the persistent kernel and the launch counts: ms5 / launches5 = [step_kernel, K1a rollout
 * (either kernel), K1b stage (or fused K1b+K1c), K1c adjoint (unfused launches only), solo_kernel];
 * solo_agents = agents that finished in the persistent wave-per-agent kernel.  Any pointer may be NULL. */
int mpc_last_kernel_profile(mpc_handle *h, double *ms5, int64_t *launches5, int64_t *solo_agents);
/* profile mode: persistent-kernel time of the last solve -- summed over the sub-batch groups (their launches
 * overlap in time: the sum can exceed the solve) and the longest single launch (the tail a blocking solve waits for) */
int mpc_last_solo_ms(mpc_handle *h, double *sum_ms, double *longest_ms);
/* A sub-batch group whose round holds at most `max_requests` evaluation requests leaves the rounds
 * and finishes in the persistent wave-per-agent kernel, and a batch of at most `max_requests` agents runs in
 * it from the start (0 = rounds only).  Defaults (measured): switch at 1024 requests (kinematic
 * model) / 128 (Pacejka model, whose persistent-kernel waves take a whole SIMD each); whole
 * batches up to 4096 agents (kinematic, N <= 32) / 1024 (otherwise); environment
 * MPC_SOLO_MAX (both) and MPC_SOLO_ALL (the batch bound alone).  Results do not depend on it. */
int mpc_set_solo_max(mpc_handle *h, int max_requests);
/* Two more per-launch kernel choosers, environment only (read at mpc_create; results do not depend on them):
 *   MPC_PAC_QUAD_MAX  Pacejka model: requests bound of a round up to which K1a runs four lanes per request
 *                     (rollout_quad_kernel); above it one thread per request (default 24576: the four-lane kernel
 *                     shortens a lone wave's chain, the thread kernel executes 2.2x fewer instructions).
 *   MPC_CHAIN_MIN     requests bound of a group's round from which the step launch carries the thread-per-agent
 *                     blocks that serve agents waiting for a trial point's gradient (default 24576: full rounds of
 *                     groups of more than 12288 agents; kinematic model only unless this variable is set);
 *                     MPC_NO_CHAIN = never. */
/* sub-batch pipelining: the batch is split into `groups` contiguous ranges whose rounds run on
 * separate HIP streams (0 = automatic: 4 from 49152 agents when the runtime runs five streams side by
 * side -- measured at mpc_create, see mpc_stream_concurrency --, 3 from 24576, 2 from 16384, else 1; at
 * most 8; never fewer than 1024 agents per group).  Results do not depend on it. */
int mpc_set_groups(mpc_handle *h, int groups);
/* streams = how many of this process's streams the HIP runtime runs side by side, as far as the solver
 * cares: 5 (five or more) or 4 (fewer).  Measured when the handle is created (an idling kernel on the null
 * stream and on four streams of the handle), NOT read from GPU_MAX_HW_QUEUES: the runtime reads that
 * variable once, when it initialises, and 4 is its default -- a C caller that wants four groups exports
 * GPU_MAX_HW_QUEUES >= 5 before its first HIP call (the Python package does so at import).
 * groups_last = sub-batch groups of the last solve.  Either pointer may be NULL. */
int mpc_stream_concurrency(mpc_handle *h, int *streams, int *groups_last);
/* on != 0 (default): a failed inner solve that the outer loop backtracks over without constraints is
 * replayed from a memo -- same outcome, same counted statistics, no evaluations executed -- instead of
 * being recomputed; 0 (or the environment MPC_NO_MEMO at mpc_create): recomputed, as the reference walks.
 * Controls, multipliers and all MPC_NSTATS statistics are bit-identical either way. */
int mpc_set_memo(mpc_handle *h, int on);
/* test aid: at most `rounds` rounds (persistent kernel: evaluations per agent) per solve; a solve that
 * needs more returns MPC_E_LIMIT with the agents it could not finish left as they are (0 = the built-in
 * guard alone, which no valid solve reaches) */
int mpc_set_round_limit(mpc_handle *h, int64_t rounds);
/* Wall-clock bound of a solve's host side (default 300 s; environment MPC_POLL_TIMEOUT_S at mpc_create).  The
 * round loop of mpc_solve_batch is host code that polls device counters: when NO polled window has completed
 * for `seconds`, or one of the blocking waits behind the loop has lasted that long, the call returns MPC_E_HIP
 * ("wall-clock bound ... expired") instead of waiting for ever on a device that does not answer.  It does NOT
 * synchronise the device on that path (that would be the same hang): kernels of the solve may still be queued and
 * the caller's U / lambda / stats buffers are in use until the caller has synchronised the device itself.  A
 * valid solve never comes near the default (a 65 536-agent solve completes a window every ~3 ms). */
int mpc_set_poll_timeout(mpc_handle *h, double seconds);
/* test aid: queues the library's idling kernel (one wavefront sleeping for `microseconds` of the device wall
 * clock, at most 30 s) on `stream` */
int mpc_debug_spin(mpc_handle *h, double microseconds, void *stream);
/* diagnostic: the per-agent solver records as the last solve left them, decoded to plain doubles, copied to the
 * HOST array host_out [B][MPC_NREC] (B <= the last solve's batch); mpc_debug_record_names() = the comma-separated
 * names of the slots in use (step sizes L / gamma, accepted line-search step tau (halved), |J|, history fill,
 * evaluation counters ...).  After a solve stopped by max_total_inner = k the records hold the solver state after
 * k inner iterations: what tests/test_gpu_parity.py::test_iterate_prefix_parity and tools/dev/first_divergence.py
 * compare with the oracle's per-iteration trace.  Synchronises the device. */
int mpc_debug_records(mpc_handle *h, int B, double *host_out);
const char *mpc_debug_record_names(void);
/* on != 0: bracket every kernel of mpc_solve_batch with HIP events on the solve's stream so that
 * mpc_last_solve_info reports eval_ms / step_ms (also enabled by the environment MPC_PROFILE=1) */
int mpc_set_profile(mpc_handle *h, int on);

#ifdef __cplusplus
}
#endif
#endif
