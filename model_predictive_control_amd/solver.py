"""BatchedMPC: torch-tensor front end of the C-ABI (include/mpc_hip.h).

Holds no arithmetic of its own: it checks shapes/dtypes/devices on the host (a
wrong shape handed to a hand-written kernel is a GPU fault) and forwards raw
device pointers plus the current HIP stream.
"""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedMPC:
    """One handle = one GPU.  All tensors are float64, contiguous, on `device`."""

    def __init__(self, cfg, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedMPC needs a HIP device (no CPU fallback exists)")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("device must be a HIP (cuda) device")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.cfg = cfg
        self.N = int(cfg.N)
        self.S = int(cfg.S)
        self.nx = self.lib.mpc_nx(C.byref(cfg))
        self.n = 2 * self.N
        self.m = self.lib.mpc_m(C.byref(cfg))
        self.M = int(cfg.lbfgs_memory)
        h = C.c_void_p()
        _lib.check(self.lib.mpc_create(C.byref(cfg), idx, C.byref(h)))
        self._h = h
        self._pending = False       # an asynchronous solve is in flight: the worker thread owns the handle
        self._cl_key, self._cl_keep = None, None   # the centerline table the search tables were last built for
        import os
        # nearest-point search of K1b: 2 grid of index ranges (default), 1 block boxes, 0 the full scan
        self._nearest_blocks = 0 if "MPC_NEAREST_SCAN" in os.environ else 1 if "MPC_NEAREST_BLOCKS" in os.environ else 2

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _free(self):
        """Refuse a call on an engine whose asynchronous solve has not been collected -- before anything
        touches the handle (the library refuses too; its tables and workspace belong to the worker)."""
        if self._pending:
            raise _lib.MpcError("a solve of this engine is in flight: call the function solve_async returned first")

    def _chk(self, t, shape, name, dtype=torch.float64):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name}: expected a torch.Tensor")
        if t.dtype != dtype:
            raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
        if t.device != self.device:
            raise ValueError(f"{name}: expected device {self.device}, got {t.device}")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        if not t.is_contiguous():
            raise ValueError(f"{name}: must be contiguous")
        return t

    def _centerline(self, cl, cl_index, B):
        self._free()
        if cl.dim() == 1:
            cl = cl.unsqueeze(0)
        if cl.dim() != 2 or cl.shape[1] != 2 * self.S:
            raise ValueError(f"centerline: expected [C, {2 * self.S}] (flat x.. then y..), "
                             f"got {tuple(cl.shape)}")
        self._chk(cl, cl.shape, "centerline")
        if cl_index is not None:
            self._chk(cl_index, (B,), "cl_index", torch.int32)
            if B and (int(cl_index.min()) < 0 or int(cl_index.max()) >= cl.shape[0]):
                raise ValueError("cl_index out of range")
        if self._nearest_blocks:
            # f-2: the tables of the pruned nearest-point searches for this centerline table (three small
            # kernels: ~30 us for one row, but 65 536 cells x 2 (S - 1) points and 256 KB of cells PER ROW for a
            # table of many rows).
            # A ONE-ROW table (the shared centerline of main.py) is rebuilt on every call: the caller may have
            # refreshed it in place by any route -- `cl.data.copy_()`, DLPack, a raw kernel -- and no host-side
            # key sees all of them; 30 us on a solve of milliseconds.
            # A table of SEVERAL rows is rebuilt only when it is another table or its contents have changed: the
            # key holds the address, the shape, torch's in-place write counter, the stream (the tables are built by
            # kernels on the caller's current stream, and a call on another stream must not read them before those
            # kernels have run) AND a checksum of the table's bits computed on the device at every call (one
            # reduction over C x 2S words, ~20 us, against ~1 ms of table building per 64 rows) -- a write behind
            # torch's back (`.data`, DLPack, a raw kernel) changes the checksum.  The engine keeps the tensor
            # alive meanwhile, so its address cannot be handed to another table.
            if cl.shape[0] == 1:
                self._cl_key, self._cl_keep = None, None
                _lib.check(self.lib.mpc_centerline_blocks(self._h, _ptr(cl), 1, self._stream()))
            else:
                key = (cl.data_ptr(), tuple(cl.shape), cl._version, self._nearest_blocks,
                       torch.cuda.current_stream(self.device).cuda_stream, int(cl.view(torch.int64).sum().item()))
                if key != self._cl_key:
                    self._cl_key, self._cl_keep = None, None
                    _lib.check(self.lib.mpc_centerline_blocks(self._h, _ptr(cl), int(cl.shape[0]), self._stream()))
                    self._cl_key, self._cl_keep = key, cl
        return cl

    def invalidate_centerline_tables(self):
        """Forget the nearest-point search tables: the next call rebuilds them for the table it is given."""
        self._cl_key, self._cl_keep = None, None

    def _empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ model layer
    def rhs(self, x, u):
        """a-1: continuous dynamics f(x, u) (car_dynamics.py:93-132 / dynamics.py:144-173)."""
        B = x.shape[0]
        self._chk(x, (B, self.nx), "x"); self._chk(u, (B, 2), "u")
        dx = self._empty(B, self.nx)
        _lib.check(self.lib.mpc_rhs(self._h, B, _ptr(x), _ptr(u), _ptr(dx), self._stream()))
        return dx

    def rollout(self, x0, U):
        """a-3: X[B, Nsim, nx] = x_1..x_Nsim (car_dynamics.py:159-166); U is [B, 2*Nsim]."""
        B = x0.shape[0]
        self._chk(x0, (B, self.nx), "x0")
        if U.dim() != 2 or U.shape[0] != B or U.shape[1] % 2:
            raise ValueError("U: expected [B, 2*Nsim]")
        self._chk(U, U.shape, "U")
        Nsim = U.shape[1] // 2
        X = self._empty(B, Nsim, self.nx)
        _lib.check(self.lib.mpc_rollout(self._h, B, Nsim, _ptr(x0), _ptr(U), _ptr(X), self._stream()))
        return X

    def stage_errors(self, pose, centerline, cl_index=None):
        """a-4/a-5: (err[B,3] = [cte, heading_error, pos_error], idx[B])."""
        B = pose.shape[0]
        self._chk(pose, (B, 3), "pose")
        cl = self._centerline(centerline, cl_index, B)
        err = self._empty(B, 3)
        idx = self._empty(B, dtype=torch.int32)
        _lib.check(self.lib.mpc_stage_errors(self._h, B, _ptr(pose), _ptr(cl), _ptr(cl_index),
                                             _ptr(err), _ptr(idx), self._stream()))
        return err, idx

    def stage_cost(self, x, u, centerline, cl_index=None):
        """a-6: L[B] (car_dynamics.py:252-258)."""
        B = x.shape[0]
        self._chk(x, (B, self.nx), "x"); self._chk(u, (B, 2), "u")
        cl = self._centerline(centerline, cl_index, B)
        out = self._empty(B)
        _lib.check(self.lib.mpc_stage_cost(self._h, B, _ptr(x), _ptr(u), _ptr(cl), _ptr(cl_index),
                                           _ptr(out), self._stream()))
        return out

    def eval_cost_grad(self, x0, centerline, U, y=None, Sigma=None, cl_index=None, want_grad=True, wave=False):
        """K1: psi[B], grad[B, 2N] (or None), yhat[B, m] (or None).  wave=True: the wave-per-agent
        evaluation of the persistent solve kernel instead of the K1a/K1b/K1c launches (same bits)."""
        B = x0.shape[0]
        self._chk(x0, (B, self.nx), "x0"); self._chk(U, (B, self.n), "U")
        cl = self._centerline(centerline, cl_index, B)
        if self.m:
            if y is None or Sigma is None:
                raise ValueError("y and Sigma are required when the problem has constraints")
            self._chk(y, (B, self.m), "y"); self._chk(Sigma, (B, self.m), "Sigma")
        psi = self._empty(B)
        grad = self._empty(B, self.n) if want_grad else None
        yhat = self._empty(B, self.m) if self.m else None
        fn = self.lib.mpc_eval_cost_grad_wave if wave else self.lib.mpc_eval_cost_grad
        _lib.check(fn(self._h, B, _ptr(x0), _ptr(cl), _ptr(cl_index), _ptr(U), _ptr(y), _ptr(Sigma), _ptr(psi),
                      _ptr(grad), _ptr(yhat), self._stream()))
        return psi, grad, yhat

    # ------------------------------------------------------------------ solver pieces
    def prox_step(self, x, grad, gamma):
        """K2: xhat, p, [||p||^2, grad'p]."""
        B = x.shape[0]
        self._chk(x, (B, self.n), "x"); self._chk(grad, (B, self.n), "grad"); self._chk(gamma, (B,), "gamma")
        xhat, p, out = self._empty(B, self.n), self._empty(B, self.n), self._empty(B, 2)
        _lib.check(self.lib.mpc_prox_step(self._h, B, _ptr(x), _ptr(grad), _ptr(gamma), _ptr(xhat),
                                          _ptr(p), _ptr(out), self._stream()))
        return xhat, p, out

    def lbfgs_apply(self, S, Y, idx, full, mask, q):
        """K3: masked two-loop on q (copied); returns (q_out, ok)."""
        B = q.shape[0]
        self._chk(S, (B, self.M, self.n), "S"); self._chk(Y, (B, self.M, self.n), "Y")
        self._chk(idx, (B,), "idx", torch.int32); self._chk(full, (B,), "full", torch.int32)
        self._chk(mask, (B, self.n), "mask"); self._chk(q, (B, self.n), "q")
        if B and (int(idx.min()) < 0 or int(idx.max()) >= self.M):
            raise ValueError("idx out of range")
        qo = q.clone()
        ok = self._empty(B, dtype=torch.int32)
        _lib.check(self.lib.mpc_lbfgs_apply(self._h, B, _ptr(S), _ptr(Y), _ptr(idx), _ptr(full), _ptr(mask),
                                            _ptr(qo), _ptr(ok), self._stream()))
        return qo, ok

    # ------------------------------------------------------------------ the solve
    def solve(self, x0, centerline, U, lam=None, cl_index=None, inplace=False):
        """a-8..a-13: returns (U*, lambda*, stats[B, 8]); warm start from U / lam."""
        B = x0.shape[0]
        self._chk(x0, (B, self.nx), "x0"); self._chk(U, (B, self.n), "U")
        cl = self._centerline(centerline, cl_index, B)
        if not inplace:
            U = U.clone()
        if self.m:
            lam = torch.zeros(B, self.m, dtype=torch.float64, device=self.device) if lam is None \
                else (lam if inplace else lam.clone())
            self._chk(lam, (B, self.m), "lam")
        else:
            lam = None
        stats = self._empty(B, _lib.NSTATS)
        _lib.check(self.lib.mpc_solve_batch(self._h, B, _ptr(x0), _ptr(cl), _ptr(cl_index), _ptr(U),
                                            _ptr(lam), _ptr(stats), self._stream()))
        return U, lam, stats

    def solve_async(self, x0, centerline, U, lam=None, cl_index=None):
        """The same solve without holding the caller's thread (mpc_solve_batch_async): returns a function;
        calling it waits for the solve (mpc_solve_wait) and returns (U*, lambda*, stats).  One solve in
        flight per engine; no other call on the engine in between."""
        self._free()
        B = x0.shape[0]
        self._chk(x0, (B, self.nx), "x0"); self._chk(U, (B, self.n), "U")
        cl = self._centerline(centerline, cl_index, B)
        U = U.clone()
        if self.m:
            lam = torch.zeros(B, self.m, dtype=torch.float64, device=self.device) if lam is None else lam.clone()
            self._chk(lam, (B, self.m), "lam")
        else:
            lam = None
        stats = self._empty(B, _lib.NSTATS)
        keep = (x0, cl, cl_index, U, lam, stats)      # the buffers stay alive until the wait
        _lib.check(self.lib.mpc_solve_batch_async(self._h, B, _ptr(x0), _ptr(cl), _ptr(cl_index), _ptr(U),
                                                  _ptr(lam), _ptr(stats), self._stream()))
        self._pending = True

        def wait():
            try:
                _lib.check(self.lib.mpc_solve_wait(self._h))
            finally:
                self._pending = False
            return keep[3], keep[4], keep[5]
        return wait

    def closed_loop(self, x, centerline, U, T, lam=None, cl_index=None, shift=False):
        """f-1 (main.py:121-146) for B agents: returns (x_T, U, lam, traj_x[B,T,nx], traj_u[B,T,2],
        failures[B], stats of the last solve)."""
        B = x.shape[0]
        self._chk(x, (B, self.nx), "x"); self._chk(U, (B, self.n), "U")
        cl = self._centerline(centerline, cl_index, B)
        x, U = x.clone(), U.clone()
        if self.m:
            lam = torch.zeros(B, self.m, dtype=torch.float64, device=self.device) if lam is None \
                else lam.clone()
        else:
            lam = None
        tx, tu = self._empty(B, T, self.nx), self._empty(B, T, 2)
        fails = torch.zeros(B, dtype=torch.int32, device=self.device)
        stats = self._empty(B, _lib.NSTATS)
        _lib.check(self.lib.mpc_closed_loop(self._h, B, int(T), int(bool(shift)), _ptr(x), _ptr(cl),
                                            _ptr(cl_index), _ptr(U), _ptr(lam), _ptr(tx), _ptr(tu),
                                            _ptr(fails), _ptr(stats), self._stream()))
        return x, U, lam, tx, tu, fails, stats

    def lane_payoff(self, ego, cars, ncars, params):
        """f-3: out[B, 2, 4] = target lane 1, 2 -> [total, safety, velocity, comfort] (game_theory.py:115-244)."""
        B = ego.shape[0]
        self._chk(ego, (B, 3), "ego")
        if cars.dim() != 3 or cars.shape[0] != B or cars.shape[2] != 3:
            raise ValueError("cars: expected [B, K, 3]")
        self._chk(cars, cars.shape, "cars"); self._chk(ncars, (B,), "ncars", torch.int32)
        K = cars.shape[1]
        if B and (int(ncars.min()) < 0 or int(ncars.max()) > K):
            raise ValueError("ncars out of range")
        pa = (C.c_double * 15)(*[float(v) for v in params])
        out = self._empty(B, 2, 4)
        _lib.check(self.lib.mpc_lane_payoff(self._h, B, K, pa, _ptr(ego), _ptr(cars), _ptr(ncars), _ptr(out),
                                            self._stream()))
        return out

    def math_probe(self, op, a, b=None):
        """Device math used by the kernels (test aid): op 0 sin, 1 cos, 2 atan, 3 atan2(a, b), 4 tan."""
        n = a.shape[0]
        self._chk(a, (n,), "a")
        if b is not None:
            self._chk(b, (n,), "b")
        out = self._empty(n)
        _lib.check(self.lib.mpc_math_probe(self._h, n, int(op), _ptr(a), _ptr(b), _ptr(out), self._stream()))
        return out

    def set_groups(self, groups):
        """Sub-batch pipelining over HIP streams (0 = automatic)."""
        _lib.check(self.lib.mpc_set_groups(self._h, int(groups)))

    def set_nearest_blocks(self, on=True):
        """Nearest-point search of K1b: 0 / False the full 98-candidate scan, 1 / True the block-pruned
        search, 2 the grid of index ranges -- the same index whichever runs."""
        self._nearest_blocks = int(on)
        _lib.check(self.lib.mpc_set_nearest_blocks(self._h, int(on)))

    def set_solo_max(self, max_requests):
        """Requests per round up to which a group finishes in the persistent wave-per-agent kernel (0 = off)."""
        _lib.check(self.lib.mpc_set_solo_max(self._h, int(max_requests)))

    def set_memo(self, on=True):
        """Failed inner solves that the outer loop backtracks over without constraints are replayed from a memo
        (default) or recomputed (False): same controls, multipliers and statistics, fewer evaluations executed."""
        _lib.check(self.lib.mpc_set_memo(self._h, int(bool(on))))

    def set_round_limit(self, rounds):
        """Test aid: cap on the rounds (persistent kernel: trips per agent) of a solve; a solve that does not
        finish inside it returns MPC_E_LIMIT.  0 = the built-in guard alone."""
        _lib.check(self.lib.mpc_set_round_limit(self._h, int(rounds)))

    def set_poll_timeout(self, seconds):
        """Wall-clock bound of a solve's host-side waits (default 300 s): a solve whose device stops answering
        raises MpcError (MPC_E_HIP) instead of blocking for ever; the device is NOT synchronised on that path."""
        _lib.check(self.lib.mpc_set_poll_timeout(self._h, float(seconds)))

    def debug_spin(self, microseconds):
        """Test aid: queue the library's idling kernel on the current stream for `microseconds`."""
        _lib.check(self.lib.mpc_debug_spin(self._h, float(microseconds), self._stream()))

    def debug_records(self, B):
        """Diagnostic: {name: array[B]} of the per-agent solver records as the last solve left them."""
        import numpy as np
        self._free()
        out = np.empty((int(B), _lib.NREC))
        _lib.check(self.lib.mpc_debug_records(self._h, int(B), C.c_void_p(out.ctypes.data)))
        names = self.lib.mpc_debug_record_names().decode().split(",")
        return {nm: out[:, i] for i, nm in enumerate(names)}

    def stream_concurrency(self):
        """(streams the HIP runtime runs side by side for this process -- 5 means five or more, measured once
        per process and device --, sub-batch groups of the last solve)."""
        self._free()
        a, b = C.c_int(), C.c_int()
        _lib.check(self.lib.mpc_stream_concurrency(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_profile(self, on=True):
        _lib.check(self.lib.mpc_set_profile(self._h, int(bool(on))))

    def last_solve_info(self):
        self._free()
        r, g, c = C.c_int64(), C.c_int64(), C.c_int64()
        e, s = C.c_double(), C.c_double()
        _lib.check(self.lib.mpc_last_solve_info(self._h, C.byref(r), C.byref(g), C.byref(c),
                                                C.byref(e), C.byref(s)))
        lm, lr = C.c_double(), C.c_int64()
        _lib.check(self.lib.mpc_last_solve_info2(self._h, C.byref(lm), C.byref(lr)))
        si, su = C.c_int64(), C.c_int64()
        _lib.check(self.lib.mpc_last_speculation(self._h, C.byref(si), C.byref(su)))
        le, lh = C.c_int64(), C.c_int64()
        _lib.check(self.lib.mpc_last_lookahead(self._h, C.byref(le), C.byref(lh)))
        km = (C.c_double * 5)()
        kl = (C.c_int64 * 5)()
        sa = C.c_int64()
        _lib.check(self.lib.mpc_last_kernel_profile(self._h, km, kl, C.byref(sa)))
        ssum, slong = C.c_double(), C.c_double()
        _lib.check(self.lib.mpc_last_solo_ms(self._h, C.byref(ssum), C.byref(slong)))
        names = ("step", "rollout", "stage", "adjoint", "solo")
        return {"kernel_ms": {k: km[i] for i, k in enumerate(names)},
                "launches": {k: int(kl[i]) for i, k in enumerate(names)}, "solo_agents": int(sa.value),
                "rounds": r.value, "evals_grad": g.value, "evals_cost": c.value,
                "eval_ms": e.value, "step_ms": s.value, "launch_pairs": int(lm.value), "lbfgs_rows": lr.value,
                "spec_issued": si.value, "spec_used": su.value, "lookahead_evals": le.value, "lookahead_hits": lh.value, "groups": self.stream_concurrency()[1],
                "solo_longest_ms": slong.value}
