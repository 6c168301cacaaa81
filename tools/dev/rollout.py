"""Development script (not a pytest test): latency of the state rollout alone."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp

dev = torch.device('cuda:0')
model = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = 20
eng = mp.BatchedMPC(mp.default_config(model, N), dev)
for B in (64, 4096, 65536):
    rng = np.random.default_rng(0)
    nx = 4 if model == 0 else 6
    x = np.zeros((B, nx)); x[:, 0] = rng.uniform(0, 5, B); x[:, 2] = rng.uniform(-.3, .3, B); x[:, 3] = rng.uniform(.3, 1.5, B)
    U = np.tile([0.5, 0.1], (B, N)) + rng.uniform(-.1, .1, (B, 2 * N))
    X0 = torch.tensor(x, device=dev); Ut = torch.tensor(U, device=dev)
    for tag in ("in range", "one lane out of range"):
        if tag != "in range":
            x2 = x.copy(); x2[::64, 3] = 80.0      # a huge speed in one lane of every wave
            X0 = torch.tensor(x2, device=dev)
        eng.rollout(X0, Ut); torch.cuda.synchronize()
        t = time.time()
        for _ in range(20): eng.rollout(X0, Ut)
        torch.cuda.synchronize()
        print("model %d B %6d %-22s %.1f us per rollout launch" % (model, B, tag, (time.time() - t) / 20 * 1e6))
