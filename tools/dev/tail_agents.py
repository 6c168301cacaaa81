"""Development script: who the slowest agents of a batch are (evaluations, iterations, evaluations per iteration)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
import bench
dev = torch.device("cuda:0")
model = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 12 if model == 1 else 20
B = 65536
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
X0 = torch.tensor(bench.synthetic_states(model, 0, B), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(model, N), dev)
U, _, st = eng.solve(X0, cl, U0)
st = st.cpu().numpy()
ev, it = st[:, 7], st[:, 2]
order = np.argsort(-ev)
print("evals: mean %.1f median %.0f p99 %.0f p99.9 %.0f max %.0f; iterations mean %.1f; evals/iteration mean %.2f" %
      (ev.mean(), np.median(ev), np.percentile(ev, 99), np.percentile(ev, 99.9), ev.max(), it.mean(), (ev / np.maximum(it, 1)).mean()))
for a in order[:12]:
    print("agent %6d evals %5.0f iterations %4.0f evals/iter %.2f status %.0f outer %.0f x0 %s" % (a, ev[a], it[a], ev[a] / max(it[a], 1), st[a, 0], st[a, 1], np.round(X0[a].cpu().numpy(), 3)))
h = np.histogram(ev, bins=[0, 200, 300, 400, 600, 800, 1000, 1500, 2000, 4000])
print("histogram of evaluations:", list(zip(h[1][1:].astype(int), h[0])))
