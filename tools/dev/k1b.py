"""Development script: K1 alone (65 536 gradient requests, then 65 536 cost requests) for a kernel trace:
   rocprofv3 --kernel-trace --stats -- python3 tools/dev/k1b.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
import bench
dev = torch.device("cuda:0")
N, B = 20, 65536
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
rng = np.random.default_rng(0)
U0 = torch.tensor(np.stack([rng.uniform(0.2, 1.0, (B, N)), rng.uniform(-0.3, 0.3, (B, N))], 2).reshape(B, 2 * N), device=dev)
eng = mp.BatchedMPC(mp.default_config(0, N), dev)
for wg in (True, False):
    for _ in range(20):
        eng.eval_cost_grad(X0, cl, U0, want_grad=wg)
torch.cuda.synchronize()
