"""Development script: the HIP state machine of ONE agent stopped after r rounds (mpc_set_round_limit), r = r0 .. r1:
phase and line-search scalars of its record at every evaluation -- a per-evaluation trace without a trace build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp
agent, r0, r1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
model, N = 1, 12
dev = torch.device("cuda:0")
cl = bench.straight_centerline()
X0 = bench.synthetic_states(model, 0, agent + 1)[agent:agent + 1]
U0 = np.tile([1.0, 0.0], (1, N))
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
PH = {0: "DONE", 1: "W_INIT_H", 2: "W_INIT_X", 3: "W_DL", 4: "W_HEUR", 5: "W_HESS", 6: "W_LS_G", 7: "W_LS_C"}
eng = mp.BatchedMPC(mp.default_config(model, N), dev)
eng.set_solo_max(0)
os.environ["MPC_NO_SPEC"] = os.environ.get("MPC_NO_SPEC", "")
for r in range(r0, r1 + 1):
    eng.set_round_limit(r)
    try:
        eng.solve(T(X0), T(cl), T(U0))
    except mp.MpcError:
        pass
    torch.cuda.synchronize()
    d = eng.debug_records(1)
    g = lambda k: d[k][0]
    print("round %3d phase %-8s k %2d tot %3d tau %-9g fb %d psie % .9e psin % .9e psixhn % .9e gpn % .6e ppn % .6e Ln %.6e L %.6e nevals %d spec %d"
          % (r, PH.get(int(g("phase")) & 63, str(g("phase"))), g("k"), g("inner_tot"), g("tau"), g("fallback"), g("psie"), g("psin"), g("psixhn"), g("gpn"), g("ppn"),
             g("Ln"), g("L"), g("nevals"), g("spec")))
