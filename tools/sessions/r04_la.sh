#!/bin/bash
R=$GRAFT_REPO_ROOT
TAG=${1:-r04g}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lookahead or persistent_kernel or memo or full_size_pacejka or solve_matches or edge_cases or closed_loop" > $OUT/tests_la.log 2>&1
rc=$?; tail -15 $OUT/tests_la.log; echo "tests rc $rc"
[ $rc -ne 0 ] && exit 4
for V in "" "MPC_NO_LOOKAHEAD=1" "" "MPC_NO_LOOKAHEAD=1"; do
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
eng.set_profile(True); eng.solve(X, cl, U0); k = eng.last_solve_info()
print("[%s] Pacejka 65536: %.1f ms %.0f solves/s rounds %d solo_agents %d longest solo %.1f ms lookahead evals %d hits %d sha %s"
      % (os.environ.get("MPC_NO_LOOKAHEAD", "la on"), sec * 1e3, B / sec, inf["rounds"], inf["solo_agents"], k["solo_longest_ms"],
         inf["lookahead_evals"], inf["lookahead_hits"], hashlib.sha256(np.ascontiguousarray(U.cpu().numpy()).tobytes()).hexdigest()[:12]), flush=True)
PY
done
