import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
import bench
dev = torch.device("cuda:0")
B, N = 65536, 20
cfg = mp.default_config(0, N, max_total_inner=600)
eng = mp.BatchedMPC(cfg, dev)
X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
U, _, st = eng.solve(X0, cl, U0)
st = st.cpu().numpy(); ev = st[:, 7]; it = st[:, 2]
print("rounds", eng.last_solve_info()["rounds"])
for q in (0.5, 0.9, 0.99, 0.999, 0.9999): print("evals q%.4f = %.0f" % (q, np.quantile(ev, q)))
print("agents with evals >", {t: int((ev > t).sum()) for t in (500, 550, 600, 650, 700, 750, 800)})
top = np.argsort(-ev)[:8]
for a in top: print("agent", a, "evals", ev[a], "iters", it[a], "evals/iter %.2f" % (ev[a] / it[a]), "outer", st[a, 1], "x0", np.round(X0[a].cpu().numpy(), 3))
