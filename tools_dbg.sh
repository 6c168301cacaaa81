#!/bin/bash
# timing-only experiments (MPC_DBG bits make results wrong): one-stream kernel times of a single solve
R=$GRAFT_REPO_ROOT
cd $R
for V in "$@"; do
env $V MPC_GROUPS=1 timeout -k 10 120 python - "$V" <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench, model_predictive_control_amd as mp
dev = torch.device("cuda:0"); N, B = 20, 65536
X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(0, N, max_total_inner=120), dev)   # bounded: wrong results must not run away
eng.solve(X0, cl, U0); eng.set_profile(True); eng.solve(X0, cl, U0)
i = eng.last_solve_info()
print("[%s]" % sys.argv[1], {k: round(v, 1) for k, v in i["kernel_ms"].items()}, "rounds", i["rounds"], flush=True)
PY
done
