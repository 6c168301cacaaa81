"""ctypes binding of libmpc_hip.so (the C-ABI declared in include/mpc_hip.h).

PyTorch is used for device memory, streams and torch.distributed only; every
kernel is hand-written HIP behind the C-ABI.  There is no CPU fallback: a
missing library or a missing GPU raises.
"""
import ctypes as C
import os
import subprocess

# The HIP runtime maps a process's streams to GPU_MAX_HW_QUEUES hardware queues (4 unless set).  A
# batched solve runs its sub-batch groups on streams of their own next to the caller's: with 8 queues
# the solver takes four groups (+3.5 % solves/s at 65 536 agents); 16 leave room for a second
# handle solving beside the first (mpc_solve_batch_async).  Read by the runtime when
# it initialises, i.e. at the first GPU call of the process: importing this package first is enough.
def _default_hw_queues():
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return
    try:    # too late once the runtime is up (it has read its environment): the solver then keeps to three groups
        import torch
        if torch.cuda.is_initialized():
            return
    except Exception:
        pass
    os.environ["GPU_MAX_HW_QUEUES"] = "16"


_default_hw_queues()

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPC_LIB_PATH", os.path.join(_HERE, "libmpc_hip.so"))  # override: dev experiments
_SRC = [os.path.join(_HERE, "csrc", f) for f in ("mpc_api.hip", "mpc_aux.hpp", "mpc_eval.hpp", "mpc_solver.hpp",
                                                  "mpc_device.hpp", "mpc_game.hpp", "mpc_solo.hpp")]
_HDR = os.path.join(os.path.dirname(_HERE), "include", "mpc_hip.h")

MODEL_KINEMATIC, MODEL_PACEJKA = 0, 1
WRAP_FLOOR, WRAP_FMOD, WRAP_IEEE = 0, 1, 2
CONSTR_NONE, CONSTR_STATE_SQ, CONSTR_LANE = 0, 1, 2
NSTATS = 8
ST_CONVERGED = 1

EXPORTS = [
    "mpc_default_config", "mpc_nx", "mpc_m", "mpc_create", "mpc_destroy", "mpc_last_error",
    "mpc_rhs", "mpc_rollout", "mpc_stage_errors", "mpc_stage_cost", "mpc_eval_cost_grad", "mpc_prox_step",
    "mpc_lbfgs_apply", "mpc_solve_batch", "mpc_solve_batch_async", "mpc_solve_wait", "mpc_closed_loop", "mpc_last_solve_info",
    "mpc_last_solve_info2", "mpc_math_probe", "mpc_set_groups", "mpc_last_kernel_ms", "mpc_lane_payoff",
    "mpc_set_profile", "mpc_last_speculation", "mpc_last_kernel_profile", "mpc_set_solo_max",
    "mpc_eval_cost_grad_wave", "mpc_centerline_blocks", "mpc_set_nearest_blocks", "mpc_set_memo",
    "mpc_set_round_limit", "mpc_stream_concurrency", "mpc_last_solo_ms",
    "mpc_set_poll_timeout", "mpc_debug_spin", "mpc_debug_records", "mpc_debug_record_names", "mpc_source_hash",
    "mpc_last_lookahead",
]
NREC = 64


class MpcConfig(C.Structure):
    """Mirror of `mpc_config` (include/mpc_hip.h)."""
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


def source_hash():
    """SHA-256 over the library's sources and its header, in a fixed order: what `build()` compiles into the
    library as its identity (mpc_source_hash) and what a committed profile names as the build it measured."""
    import hashlib
    h = hashlib.sha256()
    for p in _SRC + [_HDR]:
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """hipcc cross-compiles the library for gfx950 (works without a GPU)."""
    newest = max(os.path.getmtime(p) for p in _SRC + [_HDR])
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-DMPC_SOURCE_SHA256=\"%s\"" % source_hash(), "-o", LIB_PATH, _SRC[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=_HERE)
    return LIB_PATH


_lib = None


def load():
    """dlopen libmpc_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MPC hot path has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    cp = C.POINTER(MpcConfig)
    L.mpc_default_config.argtypes = [cp, ci, ci]
    L.mpc_nx.argtypes = [cp]
    L.mpc_m.argtypes = [cp]
    L.mpc_create.argtypes = [cp, ci, C.POINTER(vp)]
    L.mpc_destroy.argtypes = [vp]
    L.mpc_last_error.restype = C.c_char_p
    L.mpc_rhs.argtypes = [vp, ci, vp, vp, vp, vp]
    L.mpc_rollout.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    L.mpc_stage_errors.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
    L.mpc_stage_cost.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
    L.mpc_eval_cost_grad.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_eval_cost_grad_wave.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_prox_step.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_lbfgs_apply.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_solve_batch.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_solve_batch_async.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_solve_wait.argtypes = [vp]
    L.mpc_closed_loop.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mpc_last_solve_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double)]
    L.mpc_last_solve_info2.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.mpc_last_speculation.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.mpc_last_lookahead.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.mpc_math_probe.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    L.mpc_lane_payoff.argtypes = [vp, ci, ci, C.POINTER(C.c_double), vp, vp, vp, vp, vp]
    L.mpc_set_profile.argtypes = [vp, ci]
    L.mpc_set_groups.argtypes = [vp, ci]
    L.mpc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.mpc_last_kernel_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.mpc_set_solo_max.argtypes = [vp, ci]
    L.mpc_centerline_blocks.argtypes = [vp, vp, ci, vp]
    L.mpc_set_nearest_blocks.argtypes = [vp, ci]
    L.mpc_set_memo.argtypes = [vp, ci]
    L.mpc_set_round_limit.argtypes = [vp, C.c_int64]
    L.mpc_last_solo_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mpc_stream_concurrency.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mpc_set_poll_timeout.argtypes = [vp, C.c_double]
    L.mpc_debug_spin.argtypes = [vp, C.c_double, vp]
    L.mpc_debug_records.argtypes = [vp, ci, vp]
    L.mpc_debug_record_names.restype = C.c_char_p
    L.mpc_source_hash.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("mpc_last_error", "mpc_debug_record_names", "mpc_source_hash"):
            getattr(L, name).restype = ci
    _lib = L
    return L


def default_config(model=MODEL_PACEJKA, N=12, **overrides):
    """mpc_default_config + field overrides (arrays accept sequences)."""
    cfg = MpcConfig()
    rc = load().mpc_default_config(C.byref(cfg), int(model), int(N))
    if rc != 0:
        raise ValueError(load().mpc_last_error().decode())
    for k, v in overrides.items():
        cur = getattr(cfg, k)
        if hasattr(cur, "__len__"):
            for j, vv in enumerate(v):
                cur[j] = vv
        else:
            setattr(cfg, k, v)
    return cfg


def library_hash():
    """The source hash the RUNNING library was built from (mpc_source_hash)."""
    return load().mpc_source_hash().decode()


class MpcError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise MpcError(f"libmpc_hip error {rc}: {load().mpc_last_error().decode()}")
