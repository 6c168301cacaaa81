"""Development script: agents that do not converge on the Pacejka model (N = 12), HIP vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import model_predictive_control_amd as mp
from oracle import oracle as O
from conftest import straight_centerline, synthetic_states
np.set_printoptions(linewidth=200, precision=4)
dev = torch.device("cuda:0")
B, N = 256, 12
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
X0 = synthetic_states(1, B, seed=0); cl = straight_centerline(); U0 = np.tile([1., 0.], (B, N))
for mti in (2000, 5000):
    cfg = mp.default_config(1, N, max_total_inner=mti); ocfg = O.default_config(1, N, max_total_inner=mti)
    U, _, st = mp.BatchedMPC(cfg, dev).solve(T(X0), T(cl), T(U0)); U = U.cpu().numpy(); st = st.cpu().numpy()
    Uo, _, sto = O.solve_batch(ocfg, X0, cl, U0)
    bad = np.where((st[:, 0] != 1) | (sto[:, 0] != 1))[0]
    print("budget", mti, "non-converged:", bad, "iters mean hip %.1f orc %.1f max %d %d" % (st[:, 2].mean(), sto[:, 2].mean(), st[:, 2].max(), sto[:, 2].max()))
    for a in bad:
        print(" agent", a, "x0", X0[a]); print("   hip", st[a]); print("   orc", sto[a])
        X = O.rollout(ocfg, X0[a], U[a]); print("   min vx along the HIP solution's rollout %.3f; oracle's %.3f" % (X[:, 3].min(), O.rollout(ocfg, X0[a], Uo[a])[:, 3].min()))
top = np.argsort(-st[:, 2])[:5]
print("slowest HIP agents", top, st[top, 2], "oracle iters for them", sto[top, 2])
