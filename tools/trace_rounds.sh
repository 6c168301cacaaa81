#!/bin/bash
# per-window host trace of one blocking solve of bench.py's batch (MPC_HOST_TRACE): round period and requests over the solve
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/${1:-trace}; mkdir -p $OUT; shift
for V in "$@"; do
tag=$(echo $V | tr ' =' '__'); rm -f $OUT/trace_$tag.txt
env $V MPC_HOST_TIMING=1 MPC_HOST_TRACE=$OUT/trace_$tag.txt timeout -k 10 120 python - <<'PY' 2>&1 | grep -E "mpc host|solve" | tail -2
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench, model_predictive_control_amd as mp
dev = torch.device("cuda:0"); N, B = 20, int(os.environ.get("TRACE_B", 65536))
X0 = torch.tensor(bench.synthetic_states(0, 0, B), dtype=torch.float64, device=dev)
cl = torch.tensor(bench.straight_centerline(), dtype=torch.float64, device=dev)
U0 = torch.tensor([1.0, 0.0], dtype=torch.float64, device=dev).repeat(B, N)
eng = mp.BatchedMPC(mp.default_config(0, N), dev)
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); eng.solve(X0, cl, U0); torch.cuda.synchronize()
    print("solve %.2f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
PY
done
