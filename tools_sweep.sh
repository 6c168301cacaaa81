mkdir -p gpurun_out/r02j
for cfg in "MPC_GROUPS=2" "MPC_GROUPS=3" "MPC_GROUPS=4" "MPC_SOLO_MAX=512" "MPC_SOLO_MAX=2048" "MPC_SOLO_MAX=4096" "MPC_APB=16" "MPC_FUSED_MAX=32768" "MPC_FUSED_MAX=4096"; do
  env $cfg python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-pass > gpurun_out/r02j/b.json 2>/dev/null
  python - "$cfg" <<PY
import json, sys
d = json.load(open("gpurun_out/r02j/b.json"))
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 2), d["solver"]["rounds"], d["solver"]["solo_agents"], d["controls_sha256_first_65536"][:8], flush=True)
PY
done
