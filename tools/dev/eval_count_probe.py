"""Development script: evaluation counts of prefix solves (max_total_inner = k) of one agent, HIP variants vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp
from oracle import oracle as O
agent, k0, k1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
model, N = 1, 12
dev = torch.device("cuda:0")
cl = bench.straight_centerline()
X0 = bench.synthetic_states(model, 0, agent + 1)[agent:agent + 1]
U0 = np.tile([1.0, 0.0], (1, N))
T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
for k in range(k0, k1 + 1):
    so = O.solve_batch(O.default_config(model, N, max_total_inner=k), X0, cl, U0)[2][0]
    eng = mp.BatchedMPC(mp.default_config(model, N, max_total_inner=k), dev)
    U, _, st = eng.solve(T(X0), T(cl), T(U0)); st = st.cpu().numpy()[0]
    r = eng.debug_records(1)
    eng.set_solo_max(0)
    U2, _, st2 = eng.solve(T(X0), T(cl), T(U0)); st2 = st2.cpu().numpy()[0]
    print(k, "oracle evals", so[7], "HIP solo", st[7], "HIP rounds", st2[7], "| HIP L %.6e Ln %.6e tau %g psi %.9e ngrad %g ncost %g nspec %g used %g"
          % (r["L"][0], r["Ln"][0], r["tau"][0], r["psi"][0], r["ngrad"][0], r["ncost"][0], r["nspec"][0], r["nspec_used"][0]))
