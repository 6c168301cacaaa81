"""Development script (not a pytest test; library built with -DMPC_DEV_STAMP=0 for mpc_dev_records, MPC_LIB_PATH): what in an
agent's record at round R0 says that it will be among the last to finish?  A full solve gives the evaluations per agent;
the same solve stopped at R0 (mpc_set_round_limit) leaves the records as they stand there."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp
from model_predictive_control_amd import _lib

dev = torch.device("cuda:0")
MODEL = int(os.environ.get("TRACE_MODEL", 0))
N, B = int(os.environ.get("TRACE_N", 12 if MODEL else 20)), 65536
L = _lib.load()
X0 = torch.tensor(bench.synthetic_states(MODEL, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(MODEL, N), dev)
_, _, st = eng.solve(X0, cl, U0)
st = st.cpu().numpy()
ev_final = st[:, 7]
names = ("R_PSI R_L R_GAMMA R_PHI R_PSIXH R_PP R_GP R_TAU R_PSIN R_LN R_GAMMAN R_PSIXHN R_GPN R_PPN R_SIGPP R_EPS R_HN2 R_HFD "
         "R_GAMMA_TOP R_DELTA R_RHO R_EPS_OLD R_NE1 R_PS_EPS R_OUT_EPS R_OUT_DELTA R_PSI_OUT R_PSIE R_PHASE R_K R_LIDX R_LFULL "
         "R_NOPROG R_NJ R_OUTER R_FIRST R_INITRED R_PENRED R_INNER_TOT R_INNER_FAIL R_STATUS R_NEVALS").split()
ix = {n: i for i, n in enumerate(names)}
buf = np.empty((B, 64))
for R0 in [int(a) for a in sys.argv[1:]] or ([100, 200] if MODEL else [200, 300]):
    eng.set_round_limit(R0)
    try:
        eng.solve(X0, cl, U0)
    except Exception:
        pass
    assert L.mpc_dev_records(eng._h, buf.ctypes.data_as(C.POINTER(C.c_double))) == 0
    as_int = lambda col: (buf[:, ix[col]].view(np.int64) & 0xFFFFFFFF).astype(np.int64)
    phase, outer, inner_tot, k_in, nev = as_int("R_PHASE") & 63, as_int("R_OUTER"), as_int("R_INNER_TOT"), as_int("R_K"), as_int("R_NEVALS")
    eps, run_min = buf[:, ix["R_EPS"]], None
    active = phase != 0
    remaining = ev_final - nev
    print("round %d: %d agents still running; remaining evaluations among them: median %.0f, 99%% %.0f, max %.0f"
          % (R0, active.sum(), np.median(remaining[active]), np.percentile(remaining[active], 99), remaining[active].max()))
    top = np.argsort(-np.where(active, remaining, -1))[:64]          # the 64 agents with the most left
    feats = {"outer iteration (low)": -outer, "inner tolerance now (high)": eps, "inner iterations so far (high)": inner_tot + k_in,
             "evaluations per inner iteration so far (high)": nev / np.maximum(1, inner_tot + k_in),
             "iteration of the inner solve in progress (high)": k_in, "step size gamma (low)": -buf[:, ix["R_GAMMA"]],
             "Lipschitz estimate (high)": buf[:, ix["R_L"]]}
    for name, f in feats.items():
        f = np.where(active, f, -np.inf)
        for M in (256, 1024):
            pick = np.argsort(-f, kind="stable")[:M]
            print("   %-52s picks %4d: holds %2d of the 64 latest, the latest one: %s; Spearman with remaining %.2f"
                  % (name, M, len(set(pick) & set(top)), top[0] in pick,
                     np.corrcoef(np.argsort(np.argsort(f[active])), np.argsort(np.argsort(remaining[active])))[0, 1]))
