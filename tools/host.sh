#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for V in "$@"; do
echo "== $V"
env $V MPC_HOST_TIMING=1 timeout -k 10 120 python - <<'PY' 2>&1 | grep -E "mpc host|solve" | tail -4
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench, model_predictive_control_amd as mp
dev = torch.device("cuda:0"); N, B = 20, 65536
X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(0, N), dev)
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); eng.solve(X0, cl, U0); torch.cuda.synchronize()
    print("solve %.2f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
PY
done
