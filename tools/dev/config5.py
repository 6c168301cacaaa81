"""Development script (not a pytest test): BASELINE config 5 shape on one GPU -- P pairs of players,
iterated best response over batched solves (game_theory.TwoPlayerLaneChange)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from model_predictive_control_amd import game_theory as gt

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
game = gt.TwoPlayerLaneChange(N=20, device=dev, max_total_inner=600, max_total_evals=2000)
K = 4
gs = np.zeros((P, 2, 3)); gs[:, 0] = np.stack([rng.uniform(-5, 5, P), rng.uniform(8, 14, P), np.ones(P)], 1)
gs[:, 1] = np.stack([rng.uniform(-30, 30, P), rng.uniform(8, 16, P), rng.integers(1, 3, P)], 1)
xm = np.zeros((P, 2, 4)); xm[:, :, 0] = rng.uniform(0, 1, (P, 2)); xm[:, :, 3] = rng.uniform(.5, 1.0, (P, 2))
traffic = np.stack([rng.uniform(-60, 80, (P, K)), rng.uniform(0, 20, (P, K)), rng.integers(1, 3, (P, K))], 2).astype(float)
ntr = rng.integers(0, K + 1, P).astype(np.int32)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    out = game.play(gs, xm, traffic, ntr, rounds=4)
    torch.cuda.synchronize(); dt = time.time() - t
    st = out["stats"].cpu().numpy()
    print("config5 %d pairs: %.3f s for %d best-response rounds -> %.0f pair-rounds/s; converged %.3f; lane changes %d"
          % (P, dt, out["rounds"], P * out["rounds"] / dt, (st[:, 0] == 1).mean(), int((out["target"] != torch.tensor(gs[:, :, 2], device=dev).to(torch.int32)).sum())))
