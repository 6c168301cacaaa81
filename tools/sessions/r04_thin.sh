#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
bash tools/ab.sh r04thin "" "MPC_THIN_MAX=2048" "MPC_THIN_MAX=4096" "MPC_THIN_MAX=8192" "MPC_THIN_MAX=16384" "" "MPC_THIN_MAX=4096"
for V in "MPC_THIN_MAX=0" "MPC_THIN_MAX=512" "MPC_THIN_MAX=2048" "MPC_THIN_MAX=8192" "MPC_THIN_MAX=0" "MPC_THIN_MAX=2048"; do
env $V timeout -k 10 300 python - <<'PY'
import os, sys, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, model_predictive_control_amd as mp
dev = torch.device("cuda:0")
B, N = 65536, 12
eng = mp.BatchedMPC(mp.default_config(1, N), dev)
X = torch.tensor(bench.synthetic_states(1, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
sec, U, st, inf = bench.timed_solves(eng, X, cl, U0, dev, steps=3)
print("[%s] Pacejka 65536: %.1f ms %.0f solves/s rounds %d solo_agents %d sha %s"
      % (os.environ.get("MPC_THIN_MAX"), sec * 1e3, B / sec, inf["rounds"], inf["solo_agents"],
         hashlib.sha256(np.ascontiguousarray(U.cpu().numpy()).tobytes()).hexdigest()[:12]), flush=True)
PY
done
