#!/bin/bash
# Development: bench.py under a list of environment settings:  bash tools/sweep.sh "MPC_GROUPS=2" "MPC_SOLO_MAX=512 MPC_GROUPS=3" ...
# (SWEEP_ARGS="--model 1 --horizon 12" in the environment adds bench.py arguments)
mkdir -p gpurun_out/sweep
for cfg in "$@"; do
  env $cfg python bench.py --steps ${SWEEP_STEPS:-8} --warmup 2 --no-cpu-baseline --no-kernel-pass --no-pipeline-pass $SWEEP_ARGS > gpurun_out/sweep/b.json 2>/dev/null
  python - "$cfg" <<PY
import json, sys
d = json.load(open("gpurun_out/sweep/b.json"))
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 2), d["solver"]["rounds"], d["solver"]["solo_agents"], d["controls_sha256_first_65536"][:8], flush=True)
PY
done
