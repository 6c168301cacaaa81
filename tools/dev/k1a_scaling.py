"""Development script: K1 kernels alone at several request counts (rocprofv3 --kernel-trace --stats around it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
import bench
dev = torch.device("cuda:0")
N = 20
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
rng = np.random.default_rng(0)
eng = mp.BatchedMPC(mp.default_config(0, N), dev)
for B in [int(x) for x in sys.argv[1:]] or [4096, 16384, 32768, 65536, 131072]:
    X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
    U0 = torch.tensor(np.stack([rng.uniform(0.2, 1.0, (B, N)), rng.uniform(-0.3, 0.3, (B, N))], 2).reshape(B, 2 * N), device=dev)
    for _ in range(3): eng.eval_cost_grad(X0, cl, U0, want_grad=True)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10): eng.eval_cost_grad(X0, cl, U0, want_grad=True)
    ev[1].record(); torch.cuda.synchronize()
    print("B", B, "K1 (grid tables + K1a + K1b + K1c) per call: %.1f us" % (ev[0].elapsed_time(ev[1]) * 100), flush=True)
