/*
 * mpc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's batched-MPC hot path
 * (panagiotou23/model-predictive-control).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (model_predictive_control_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: model/cost layer is pinned against the reference's importable
 * NumPy twin (dynamics.py) through tests/golden/; the solver layer (alpaqa
 * 0.0.1 StructuredPANOCLBFGS + ALM, a pip dependency absent from
 * /root/reference, no pinned version, no golden vectors in the reference) is
 * "parity unpinned": it restates alpaqa's published algorithm and is
 * cross-checked by finite differences, scipy L-BFGS-B and KKT residuals.
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MODEL_KINEMATIC = 0, ORC_MODEL_PACEJKA = 1 };
enum { ORC_WRAP_FLOOR = 0, ORC_WRAP_FMOD = 1, ORC_WRAP_IEEE = 2 };
enum { ORC_CONSTR_NONE = 0, ORC_CONSTR_STATE_SQ = 1, ORC_CONSTR_LANE = 2 };
/* alpaqa SolverStatus, as read at controller.py:59-64 */
enum {
    ORC_ST_UNKNOWN = 0, ORC_ST_CONVERGED = 1, ORC_ST_MAXTIME = 2, ORC_ST_MAXITER = 3,
    ORC_ST_NOTFINITE = 4, ORC_ST_NOPROGRESS = 5, ORC_ST_INTERRUPTED = 6
};

typedef struct orc_config {
    int32_t model;           /* ORC_MODEL_* */
    int32_t N;               /* horizon (main.py:68 N_horiz) */
    int32_t S;               /* centerline points (main.py:70) */
    int32_t nfe;             /* RK4 finite elements per stage (car_dynamics.py:136) */
    int32_t wrap_mode;       /* ORC_WRAP_* (car_dynamics.py:168-172 ambiguity) */
    int32_t clip_inputs;     /* dynamics.py:99/:163 np.clip of the inputs */
    int32_t constr_mode;     /* ORC_CONSTR_* */
    int32_t lbfgs_memory;    /* controller.py:36 */
    int32_t max_iter;        /* PANOC max_iter, controller.py:31 */
    int32_t max_outer;       /* ALM max_iter, controller.py:45 */
    int32_t hess_heuristic;  /* hessian_step_size_heuristic, controller.py:32 */
    int32_t max_no_progress; /* alpaqa default 10 */
    double Ts;               /* car_dynamics.py:93 */
    double v_ref;            /* main.py:65 */
    double cost_w[6];        /* car_dynamics.py:230 */
    double veh[22];          /* car_dynamics.py:65-88 order, main.py:82-111 values */
    double accel, friction;  /* dynamics.py:34-35 (kinematic model only) */
    double u_lb[2], u_ub[2]; /* main.py:55-56 box C, order [d, delta] */
    double g_off[6];         /* main.py:46-51: g_i = x_i^2 - g_off[i] */
    double D_lb[6], D_ub[6]; /* bounds on g per component, tiled over stages (main.py:57) */
    double lane_halfwidth;   /* ORC_CONSTR_LANE: |signed distance| <= halfwidth */
    /* ALM (controller.py:39-46 + alpaqa 0.0.1 defaults) */
    double alm_eps, alm_delta, Sigma0, eps0, rho, Delta, theta, M, Sigma_max;
    double Delta_lower, Sigma0_lower, eps0_increase, rho_increase;
    int32_t max_num_initial_retries, max_num_retries, max_total_num_retries;
    int32_t max_total_inner; /* budget of inner iterations per solve: deterministic stand-in for the
                                wall-clock caps controller.py:30,:44 (status MAXTIME when hit) */
    int32_t max_total_evals; /* budget of psi / grad-psi evaluations per solve (0 = none): the closer
                                stand-in for a wall-clock cap -- time is spent per evaluation, and a
                                line search that backtracks to tau_min costs ~18 of them per iteration.
                                Checked where alpaqa checks the clock (the inner stop test); an inner
                                solve stopped by it hands back its iterate only if `overwrite` says so */
    /* PANOC (alpaqa 0.0.1 defaults) */
    double lip_eps, lip_delta, Lgamma_factor, L_min, L_max, tau_min, qub_tol;
} orc_config;

/* number of stats doubles per agent written by orc_solve_batch */
#define ORC_NSTATS 8
/* stats layout: [status, outer_iters, inner_iters, inner_failures, eps, delta, psi, n_evals] */

void orc_default_config(orc_config *c, int model, int N);
int  orc_nx(const orc_config *c);
int  orc_m(const orc_config *c);   /* number of general constraints m_c */

/* a-1 / a-1b: continuous RHS */
void orc_rhs(const orc_config *c, const double *x, const double *u, double *dx);
/* a-2: one discrete stage (nfe RK4 steps) */
void orc_fd(const orc_config *c, const double *x, const double *u, double *xn);
/* a-3: rollout, X is [N][nx] = x_1..x_N */
void orc_rollout(const orc_config *c, const double *x0, const double *U, double *X);
/* a-4: nearest point index (strict <, candidates 0..S-2) */
int  orc_nearest(const orc_config *c, const double *pos, const double *cl);
/* a-5: errors; out = [cte, heading_error, pos_error] */
void orc_errors(const orc_config *c, const double *pos, double phi, const double *cl, double *out);
/* a-6: stage cost */
double orc_stage_cost(const orc_config *c, const double *x, const double *u, const double *cl);
/* a-7: g(U) (m values) */
void orc_constraints(const orc_config *c, const double *x0, const double *cl, const double *U, double *g);
/* a-9: psi (and grad if grad != NULL); y,Sigma may be NULL when m == 0; yhat (m) optional out */
double orc_psi(const orc_config *c, const double *x0, const double *cl, const double *U,
               const double *y, const double *Sigma, double *grad, double *yhat);

/* a-8..a-12: full ALM + structured PANOC solve for one agent.
 * U (n) and lam (m) are warm-start in / solution out. stats: ORC_NSTATS doubles. */
void orc_solve(const orc_config *c, const double *x0, const double *cl,
               double *U, double *lam, double *stats);

/* the same, recording one row per ALM outer iteration into trace[max_rows][ORC_TRACE_COLS]:
 * [outer, eps asked, inner status, inner iterations, eps reached, ||err_z||_inf (NaN: iterate not
 * handed back), min Sigma, max Sigma, backtrack, overwrite, evaluations so far, ||lambda||_inf];
 * returns the number of rows written */
#define ORC_TRACE_COLS 12
int orc_solve_traced(const orc_config *c, const double *x0, const double *cl, double *U, double *lam,
                     double *stats, double *trace, int max_rows);

/* the same, recording one row per accepted INNER iteration into trace[max_rows][ORC_ITRACE_COLS] (study aid of
 * the HIP-vs-oracle first-divergence diagnosis): [inner iterations so far (over all inner solves), ALM outer
 * iteration, iteration of this inner solve, eps asked, tau of the accepted trial (negative: the safe prox step),
 * line-search trials, L, gamma, |J|, L-BFGS pairs held, pair accepted, psi, phi_gamma, ||p||^2, evaluations so
 * far, and the smallest margin by which this iteration decided its line-search tests / descent-lemma tests /
 * active-set memberships (absolute) / the stop test ON the iterate it produced (eps_k / eps - 1; NaN when that
 * test belongs to the next inner solve) / the step-size heuristic];
 * returns the number of iterations (rows beyond max_rows are not written) */
#define ORC_ITRACE_COLS 20
int orc_solve_itertrace(const orc_config *c, const double *x0, const double *cl, double *U, double *lam,
                        double *stats, double *trace, int max_rows);

/* study aid: the solve of one agent, recording every evaluation made during inner iteration `iteration` (counted
 * over all inner solves, from 1) into out[max_rows][n + 2] = [1 gradient / 0 cost only, psi returned, the point];
 * returns the number of rows */
int orc_solve_dump_evals(const orc_config *c, const double *x0, const double *cl, double *U, double *lam,
                         double *stats, int iteration, double *out, int max_rows);

/* study aid: from now on every psi value and gradient component the SOLVER sees (orc_solve*, not orc_psi*) is moved
 * by a random whole number of ulps in [-ulps, ulps] (generator keyed by the agent's initial state and `seed`); 0 = off
 * (default).  Models another correct implementation of the same formulae: the yardstick for HIP-vs-oracle agreement
 * of solver PATHS, which depend on the last bits of the evaluations (finite-difference Hessian-vector products
 * divide gradient differences by h ~ 3e-5).  Process-wide; not thread-safe against running solves. */
void orc_set_eval_jitter(int ulps, uint64_t seed);

/* batch: x0 [B][nx], cl table [C][2S], cl_index [B] (NULL -> all 0), U [B][n], lam [B][m] */
void orc_solve_batch(const orc_config *c, int B, const double *x0, const double *cl,
                     const int32_t *cl_index, double *U, double *lam, double *stats,
                     int nthreads);
/* y, Sigma: [B][m] (ignored when m == 0) */
void orc_psi_batch(const orc_config *c, int B, const double *x0, const double *cl,
                   const int32_t *cl_index, const double *U, const double *y, const double *Sigma,
                   double *psi, double *grad, int nthreads);
int orc_max_threads(void);
/* diagnostics of the last orc_solve on the calling thread: out[8] = [line-search trials, iterations,
 * descent-lemma doublings inside trials, doublings at the top of an iteration, longest run of doublings,
 * Hessian-vector evaluations, iterations that reached the safe step tau < tau_min, 0] */
void orc_last_ls_counters(double *out);

/* f-3: lane-change payoffs (game_theory.py:115-244).  params[15] = [L, W, l, theta_max, tlc, td, ti, tau,
 * a_max, h, Lf, q1, q2, a, b]; ego [B][3] = (x, v, lane); cars [B][K][3]; ncars [B] (<= K).
 * out [B][2][4] = for target lane 1 and 2: [total, safety, velocity, comfort] of the ego. */
void orc_lane_payoff(const double *params, int B, int K, const double *ego, const double *cars,
                     const int32_t *ncars, double *out);

#ifdef __cplusplus
}
#endif
#endif
