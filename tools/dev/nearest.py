"""Development script: timing of K1 with the block-pruned nearest-point search on / off."""
import os, sys, time
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
for rep in range(2):
    for on in (2, 1, 0):
        eng.set_nearest_blocks(on)
        for wg in (True, False):
            eng.eval_cost_grad(X0, cl, U0, want_grad=wg)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): eng.eval_cost_grad(X0, cl, U0, want_grad=wg)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
            print("blocks", on, "want_grad", wg, "K1 (3 launches, 65536 requests): %.1f us" % (dt * 1e6), flush=True)
        eng.set_solo_max(1024)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); U, _, st = eng.solve(X0, cl, U0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        import hashlib
        print("blocks", on, "solve %.2f ms -> %.0f solves/s" % (min(ts) * 1e3, B / min(ts)),
              hashlib.sha256(U.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
