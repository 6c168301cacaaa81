"""Development script: how much of the step kernel is the L-BFGS two-loop (history pairs read per solve)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
import bench
dev = torch.device("cuda:0")
N, B = 20, 65536
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(0, N), dev)
U, _, st = eng.solve(X0, cl, U0)
info = eng.last_solve_info()
it = float(st[:, 2].sum()) if st.shape[1] > 2 else 0
print({k: info[k] for k in ("rounds", "evals_grad", "evals_cost", "lbfgs_rows", "spec_issued", "spec_used")})
print("inner iterations (sum of stats col 2..):", [float(st[:, j].sum()) for j in range(st.shape[1])])
