"""ctypes binding of the CPU ORACLE (oracle/libmpc_oracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by model_predictive_control_amd/.
See oracle/mpc_oracle.h for the parity status ("parity unpinned" for the
solver layer; model layer pinned by tests/golden/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpc_oracle.so")

MODEL_KINEMATIC, MODEL_PACEJKA = 0, 1
WRAP_FLOOR, WRAP_FMOD, WRAP_IEEE = 0, 1, 2
CONSTR_NONE, CONSTR_STATE_SQ, CONSTR_LANE = 0, 1, 2
NSTATS = 8


class OrcConfig(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("N", C.c_int32), ("S", C.c_int32), ("nfe", C.c_int32),
        ("wrap_mode", C.c_int32), ("clip_inputs", C.c_int32), ("constr_mode", C.c_int32),
        ("lbfgs_memory", C.c_int32), ("max_iter", C.c_int32), ("max_outer", C.c_int32),
        ("hess_heuristic", C.c_int32), ("max_no_progress", C.c_int32),
        ("Ts", C.c_double), ("v_ref", C.c_double), ("cost_w", C.c_double * 6),
        ("veh", C.c_double * 22), ("accel", C.c_double), ("friction", C.c_double),
        ("u_lb", C.c_double * 2), ("u_ub", C.c_double * 2), ("g_off", C.c_double * 6),
        ("D_lb", C.c_double * 6), ("D_ub", C.c_double * 6), ("lane_halfwidth", C.c_double),
        ("alm_eps", C.c_double), ("alm_delta", C.c_double), ("Sigma0", C.c_double),
        ("eps0", C.c_double), ("rho", C.c_double), ("Delta", C.c_double), ("theta", C.c_double),
        ("M", C.c_double), ("Sigma_max", C.c_double), ("Delta_lower", C.c_double),
        ("Sigma0_lower", C.c_double), ("eps0_increase", C.c_double), ("rho_increase", C.c_double),
        ("max_num_initial_retries", C.c_int32), ("max_num_retries", C.c_int32),
        ("max_total_num_retries", C.c_int32), ("max_total_inner", C.c_int32),
        ("max_total_evals", C.c_int32),
        ("lip_eps", C.c_double), ("lip_delta", C.c_double), ("Lgamma_factor", C.c_double),
        ("L_min", C.c_double), ("L_max", C.c_double), ("tau_min", C.c_double),
        ("qub_tol", C.c_double),
    ]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "mpc_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmpc_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        cp = C.POINTER(OrcConfig)
        L.orc_default_config.argtypes = [cp, C.c_int, C.c_int]
        L.orc_nx.argtypes = [cp]; L.orc_nx.restype = C.c_int
        L.orc_m.argtypes = [cp]; L.orc_m.restype = C.c_int
        L.orc_rhs.argtypes = [cp, dp, dp, dp]
        L.orc_fd.argtypes = [cp, dp, dp, dp]
        L.orc_rollout.argtypes = [cp, dp, dp, dp]
        L.orc_nearest.argtypes = [cp, dp, dp]; L.orc_nearest.restype = C.c_int
        L.orc_errors.argtypes = [cp, dp, C.c_double, dp, dp]
        L.orc_stage_cost.argtypes = [cp, dp, dp, dp]; L.orc_stage_cost.restype = C.c_double
        L.orc_constraints.argtypes = [cp, dp, dp, dp, dp]
        L.orc_psi.argtypes = [cp, dp, dp, dp, dp, dp, dp, dp]; L.orc_psi.restype = C.c_double
        L.orc_solve.argtypes = [cp, dp, dp, dp, dp, dp]
        L.orc_solve_traced.argtypes = [cp, dp, dp, dp, dp, dp, dp, C.c_int]; L.orc_solve_traced.restype = C.c_int
        L.orc_solve_itertrace.argtypes = [cp, dp, dp, dp, dp, dp, dp, C.c_int]; L.orc_solve_itertrace.restype = C.c_int
        L.orc_solve_dump_evals.argtypes = [cp, dp, dp, dp, dp, dp, C.c_int, dp, C.c_int]; L.orc_solve_dump_evals.restype = C.c_int
        L.orc_set_eval_jitter.argtypes = [C.c_int, C.c_uint64]
        L.orc_solve_batch.argtypes = [cp, C.c_int, dp, dp, ip, dp, dp, dp, C.c_int]
        L.orc_psi_batch.argtypes = [cp, C.c_int, dp, dp, ip, dp, dp, dp, dp, dp, C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_lane_payoff.argtypes = [dp, C.c_int, C.c_int, dp, dp, ip, dp]
        _lib = L
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_config(model=MODEL_PACEJKA, N=12, **kw):
    cfg = OrcConfig()
    lib().orc_default_config(C.byref(cfg), model, N)
    for k, v in kw.items():
        cur = getattr(cfg, k)
        if hasattr(cur, "__len__"):
            for j, vv in enumerate(v):
                cur[j] = vv
        else:
            setattr(cfg, k, v)
    return cfg


def nx(cfg):
    return lib().orc_nx(C.byref(cfg))


def m(cfg):
    return lib().orc_m(C.byref(cfg))


def rhs(cfg, x, u):
    x, u = _f64(x), _f64(u)
    out = np.empty(nx(cfg))
    lib().orc_rhs(C.byref(cfg), _d(x), _d(u), _d(out))
    return out


def fd(cfg, x, u):
    x, u = _f64(x), _f64(u)
    out = np.empty(nx(cfg))
    lib().orc_fd(C.byref(cfg), _d(x), _d(u), _d(out))
    return out


def rollout(cfg, x0, U):
    x0, U = _f64(x0), _f64(U)
    X = np.empty((cfg.N, nx(cfg)))
    lib().orc_rollout(C.byref(cfg), _d(x0), _d(U), _d(X))
    return X


def nearest(cfg, pos, cl):
    pos, cl = _f64(pos), _f64(cl)
    return lib().orc_nearest(C.byref(cfg), _d(pos), _d(cl))


def errors(cfg, pos, phi, cl):
    pos, cl = _f64(pos), _f64(cl)
    out = np.empty(3)
    lib().orc_errors(C.byref(cfg), _d(pos), float(phi), _d(cl), _d(out))
    return out


def stage_cost(cfg, x, u, cl):
    x, u, cl = _f64(x), _f64(u), _f64(cl)
    return lib().orc_stage_cost(C.byref(cfg), _d(x), _d(u), _d(cl))


def constraints(cfg, x0, cl, U):
    x0, cl, U = _f64(x0), _f64(cl), _f64(U)
    g = np.empty(m(cfg))
    lib().orc_constraints(C.byref(cfg), _d(x0), _d(cl), _d(U), _d(g))
    return g


def psi(cfg, x0, cl, U, y=None, Sigma=None, want_grad=True):
    x0, cl, U = _f64(x0), _f64(cl), _f64(U)
    y = None if y is None else _f64(y)
    Sigma = None if Sigma is None else _f64(Sigma)
    g = np.empty(2 * cfg.N) if want_grad else None
    v = lib().orc_psi(C.byref(cfg), _d(x0), _d(cl), _d(U), _d(y), _d(Sigma), _d(g), None)
    return v, g


def solve(cfg, x0, cl, U0, lam0=None):
    x0, cl = _f64(x0), _f64(cl)
    U = _f64(U0).copy()
    mm = m(cfg)
    lam = np.zeros(max(mm, 1)) if lam0 is None else _f64(lam0).copy()
    st = np.empty(NSTATS)
    lib().orc_solve(C.byref(cfg), _d(x0), _d(cl), _d(U), _d(lam), _d(st))
    return U, lam[:mm], st


TRACE_COLS = ["outer", "eps_asked", "inner_status", "inner_iters", "eps_reached", "err_z_inf", "Sigma_min",
              "Sigma_max", "backtrack", "overwrite", "evals", "lambda_inf"]


def solve_traced(cfg, x0, cl, U0, lam0=None, max_rows=256):
    """solve() plus one trace row per ALM outer iteration (columns: TRACE_COLS)."""
    x0, cl = _f64(x0), _f64(cl)
    U = _f64(U0).copy()
    mm = m(cfg)
    lam = np.zeros(max(mm, 1)) if lam0 is None else _f64(lam0).copy()
    st = np.empty(NSTATS)
    tr = np.zeros((max_rows, len(TRACE_COLS)))
    k = lib().orc_solve_traced(C.byref(cfg), _d(x0), _d(cl), _d(U), _d(lam), _d(st), _d(tr), max_rows)
    return U, lam[:mm], st, tr[:k]


ITRACE_COLS = ["inner_total", "outer", "k", "eps_asked", "tau", "trials", "L", "gamma", "nJ", "lbfgs_pairs",
               "pair_ok", "psi", "phi", "pp", "evals", "margin_ls", "margin_dl", "margin_active", "margin_stop",
               "margin_heur"]


def solve_itertrace(cfg, x0, cl, U0, lam0=None, max_rows=6000):
    """solve() plus one trace row per accepted inner iteration (columns: ITRACE_COLS; mpc_oracle.h)."""
    x0, cl = _f64(x0), _f64(cl)
    U = _f64(U0).copy()
    mm = m(cfg)
    lam = np.zeros(max(mm, 1)) if lam0 is None else _f64(lam0).copy()
    st = np.empty(NSTATS)
    tr = np.zeros((max_rows, len(ITRACE_COLS)))
    k = lib().orc_solve_itertrace(C.byref(cfg), _d(x0), _d(cl), _d(U), _d(lam), _d(st), _d(tr), max_rows)
    return U, lam[:mm], st, tr[:min(k, max_rows)]


def solve_dump_evals(cfg, x0, cl, U0, iteration, max_rows=512):
    """Every evaluation the solve makes during inner iteration `iteration`: rows [is_gradient, psi, point...]."""
    x0, cl = _f64(x0), _f64(cl)
    U = _f64(U0).copy()
    lam = np.zeros(max(m(cfg), 1))
    st = np.empty(NSTATS)
    n = 2 * cfg.N
    out = np.zeros((max_rows, n + 2))
    k = lib().orc_solve_dump_evals(C.byref(cfg), _d(x0), _d(cl), _d(U), _d(lam), _d(st), int(iteration), _d(out), max_rows)
    return out[:k]


class eval_jitter:
    """`with eval_jitter(ulps, seed):` -- solves inside see every psi / gradient component moved by a random whole
    number of ulps in [-ulps, ulps] (mpc_oracle.h orc_set_eval_jitter): another correct implementation's evaluations."""

    def __init__(self, ulps, seed=0):
        self.ulps, self.seed = int(ulps), int(seed)

    def __enter__(self):
        lib().orc_set_eval_jitter(self.ulps, self.seed)

    def __exit__(self, *exc):
        lib().orc_set_eval_jitter(0, 0)


def solve_batch(cfg, x0, cl, U0, lam0=None, cl_index=None, nthreads=0):
    x0, cl = _f64(x0), _f64(cl)
    B = x0.shape[0]
    U = _f64(U0).copy()
    mm = m(cfg)
    lam = np.zeros((B, max(mm, 1))) if lam0 is None else _f64(lam0).copy()
    st = np.empty((B, NSTATS))
    ci = None if cl_index is None else np.ascontiguousarray(cl_index, dtype=np.int32)
    if nthreads <= 0:
        nthreads = lib().orc_max_threads()
    lib().orc_solve_batch(C.byref(cfg), B, _d(x0), _d(cl), _i(ci), _d(U), _d(lam), _d(st),
                          nthreads)
    return U, lam[:, :mm], st


def psi_batch(cfg, x0, cl, U, y=None, Sigma=None, cl_index=None, want_grad=True, nthreads=0):
    x0, cl, U = _f64(x0), _f64(cl), _f64(U)
    B = x0.shape[0]
    y = None if y is None else _f64(y)
    Sigma = None if Sigma is None else _f64(Sigma)
    p = np.empty(B)
    g = np.empty((B, 2 * cfg.N)) if want_grad else None
    ci = None if cl_index is None else np.ascontiguousarray(cl_index, dtype=np.int32)
    if nthreads <= 0:
        nthreads = lib().orc_max_threads()
    lib().orc_psi_batch(C.byref(cfg), B, _d(x0), _d(cl), _i(ci), _d(U), _d(y), _d(Sigma), _d(p),
                        _d(g), nthreads)
    return p, g


# game_theory.py:21-56 Car defaults + get_safety_distance q1, q2 + get_total_payoff a, b
LANE_PARAMS = np.array([4.2, 1.8, 3.0, 3.2 / 180 * np.pi, 5.17, 1.2, 0.15, 0.9, 7.0, 3.75, 1.0,
                        0.65, 0.35, 0.6, 0.4])


def lane_payoff(ego, cars, ncars, params=LANE_PARAMS):
    """out [B, 2, 4]: target lane 1 / 2 -> [total, safety, velocity, comfort] (game_theory.py:115-244)."""
    ego, cars = _f64(ego), _f64(cars)
    B, K = cars.shape[0], cars.shape[1]
    nc = np.ascontiguousarray(ncars, dtype=np.int32)
    out = np.empty((B, 2, 4))
    lib().orc_lane_payoff(_d(_f64(params)), B, K, _d(ego), _d(cars), _i(nc), _d(out))
    return out
