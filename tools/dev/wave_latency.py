"""Development script: latency of ONE evaluation by one wave (the persistent kernel's evaluation),
and of a whole single-agent solve, by model."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
import bench
dev = torch.device("cuda:0")
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
for model, N in ((0, 20), (1, 12), (0, 40)):
    for B in (1, 256):
        X0 = torch.tensor(bench.synthetic_states(model, 0, B), dtype=torch.float64, device=dev)
        U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
        eng = mp.BatchedMPC(mp.default_config(model, N), dev)
        for wave in (True, False):
            for wg in (True, False):
                eng.eval_cost_grad(X0, cl, U0, want_grad=wg, wave=wave)
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(200): eng.eval_cost_grad(X0, cl, U0, want_grad=wg, wave=wave)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 200
                print("model %d N %d B %d %s want_grad %-5s: %.1f us per call" % (model, N, B, "wave " if wave else "3-kernel", wg, dt * 1e6), flush=True)
        for sm in (0, 100000):
            eng.set_solo_max(sm)
            eng.solve(X0, cl, U0)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(3): U, _, st = eng.solve(X0, cl, U0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
            print("model %d N %d B %d solve solo_max %d: %.2f ms, evals mean %.0f max %.0f -> %.1f us per eval of the slowest agent" % (model, N, B, sm, dt * 1e3, float(st[:, 7].mean()), float(st[:, 7].max()), dt * 1e6 / float(st[:, 7].max())), flush=True)
