"""Development script (not a pytest test): the persistent wave-per-agent kernel against the round
path -- timing by batch size and switch threshold, and equality of the results."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
import bench

dev = torch.device("cuda:0")
model = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sizes = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 64, 1024, 4096, 65536]
thresholds = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 512, 1024, 2048, 4096]
evals = int(sys.argv[5]) if len(sys.argv) > 5 else 0
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
for B in sizes:
    X0 = torch.tensor(bench.synthetic_states(model, 0, B), dtype=torch.float64, device=dev)
    U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
    ref = None
    for th in thresholds:
        eng = mp.BatchedMPC(mp.default_config(model, N, max_total_evals=evals), dev)
        if th >= 0:            # negative: the library's defaults
            eng.set_solo_max(th)
        eng.solve(X0, cl, U0)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            U, _, st = eng.solve(X0, cl, U0)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        info = eng.last_solve_info()
        same = "-" if ref is None else str(bool(torch.equal(U, ref[0]) and torch.equal(st, ref[1])))
        if ref is None:
            ref = (U, st)
        print("B %6d solo_max %6d: %8.2f ms (min of 3) -> %9.0f solves/s  rounds %5d solo_agents %6d conv %.3f same_bits %s"
              % (B, th, min(ts) * 1e3, B / min(ts), info["rounds"], info["solo_agents"],
                 float((st[:, 0] == 1).double().mean()), same), flush=True)
