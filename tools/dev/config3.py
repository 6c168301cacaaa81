import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import model_predictive_control_amd as mp
from model_predictive_control_amd.bezier_curves import lane_change_centerlines
dev = torch.device('cuda:0')
B, N = int(sys.argv[1]), 40
cfg = mp.default_config(0, N, constr_mode=2, lane_halfwidth=0.05, max_total_inner=1000, max_total_evals=4000, Sigma0=10.0)
eng = mp.BatchedMPC(cfg, dev)
eng.set_profile(len(sys.argv) > 2)
tabs = lane_change_centerlines(S=100)
rng = np.random.default_rng(0)
x = np.stack([rng.uniform(0, 2, B), rng.uniform(-.02, .02, B), rng.uniform(-.05, .05, B), rng.uniform(.5, 1.2, B)], 1)
idx = rng.integers(0, tabs.shape[0], B).astype(np.int32)
X0 = torch.tensor(x, device=dev); cl = torch.tensor(tabs, device=dev); ci = torch.tensor(idx, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    U, lam, st = eng.solve(X0, cl, U0, cl_index=ci)
    torch.cuda.synchronize(); dt = time.time() - t
    st = st.cpu().numpy(); info = eng.last_solve_info()
    print("config3 B=%d N=40 lane constraints: %.3f s -> %.0f solves/s; status %s; iters mean %.1f; evals mean %.1f max %.0f; rounds %d"
          % (B, dt, B / dt, dict(zip(*np.unique(st[:, 0], return_counts=True))), st[:, 2].mean(), st[:, 7].mean(), st[:, 7].max(), info["rounds"]))
    if len(sys.argv) > 2: print("   kernel ms", {k: round(v, 1) for k, v in info["kernel_ms"].items()}, "lbfgs pairs", info["lbfgs_rows"], "spec", info["spec_issued"], info["spec_used"])
