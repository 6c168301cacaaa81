#!/bin/bash
# whole GPU suite (+ optionally under a switch set) and the bench line
R=$GRAFT_REPO_ROOT
TAG=${1:-r04h}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
echo "suite rc $?"; tail -6 $OUT/tests.log
if [ -n "$VARIANT" ]; then
  env $VARIANT timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/tests_variant.log 2>&1
  echo "suite under [$VARIANT] rc $?"; tail -4 $OUT/tests_variant.log
fi
timeout -k 10 400 python bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -20 $OUT/bench.err; }
python - $OUT/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("value %.0f ms %.2f flops_fraction %.3f executed %.3f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["flops_fraction"], d["roofline"]["fp64_valu"]["frac_executed"], d["roofline"]["traffic"]))
print("parity_sample", {k: d["parity_sample"][k] for k in ("max_rel_dU","frac_dU_le_1e-5","identical_path_frac")})
p=d["parity_at_1e-5"]; print("parity leg %.0f" % p["value"], {k: p["parity_sample"][k] for k in ("agents","max_rel_dU","frac_dU_le_1e-5")})
for k,v in d["secondary"].items(): print(k, "%.0f solves/s %.1f ms" % (v["value"], v["ms_per_step"]), v.get("converged_frac"), (v.get("parity_sample") or {}).get("max_rel_dU"), (v.get("cpu_baseline") or {}).get("value"), v.get("solo_kernel_share"))
print("pipelined", d.get("pipelined_two_handles",{}).get("value"))
print("sha", d["controls_sha256_first_65536"][:12], d["secondary"]["pacejka_nx6_N12"]["controls_sha256"][:12])
PY
