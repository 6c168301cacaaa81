"""Development script (not a pytest test): K1 alone (mpc_eval_cost_grad) at growing batch sizes, meant to
run under `rocprofv3 --kernel-trace`: how do the K1a / K1b+K1c launch times grow with the number of waves?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import model_predictive_control_amd as mp

dev = torch.device("cuda:0")
N = 20
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768, 49152, 65536, 98304, 131072]
Bm = max(sizes)
X0 = torch.tensor(bench.synthetic_states(0, 0, Bm), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
rng = np.random.default_rng(1)
U = torch.tensor(np.tile([1.0, 0.0], (Bm, N)) + 0.05 * rng.standard_normal((Bm, 2 * N)), dtype=torch.float64, device=dev)
eng = mp.BatchedMPC(mp.default_config(0, N), dev)
for B in sizes:
    for rep in range(4):
        eng.eval_cost_grad(X0[:B], cl, U[:B])
    torch.cuda.synchronize()
print("done", sizes)
