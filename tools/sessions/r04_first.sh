#!/bin/bash
# round 4, first GPU session: the new tests, the first-divergence study, the whole suite, the bench line
R=$GRAFT_REPO_ROOT
TAG=${1:-r04a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "iterate_prefix or in_place_centerline or wall_clock or getters_are_refused or bench_parity_tolerance" > $OUT/tests_new.log 2>&1
rc=$?; tail -25 $OUT/tests_new.log; echo "new tests rc $rc"
timeout -k 10 500 python tools/dev/first_divergence.py --out $OUT/first_divergence.txt > $OUT/first_divergence.log 2> $OUT/first_divergence.err || { echo "first_divergence failed"; tail -20 $OUT/first_divergence.err; }
tail -60 $OUT/first_divergence.log
if [ "${SKIP_SUITE:-0}" = 0 ]; then
timeout -k 10 700 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
echo "suite rc $?"; tail -8 $OUT/tests.log
fi
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -20 $OUT/bench.err; }
python - $OUT/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("value %.0f ms %.2f flops_fraction %.3f executed %.3f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["flops_fraction"], d["roofline"]["fp64_valu"]["frac_executed"], d["roofline"]["traffic"]))
print("parity_sample", {k: d["parity_sample"][k] for k in ("max_rel_dU","frac_dU_le_1e-5","identical_path_frac")})
p=d["parity_at_1e-5"]; print("parity leg %.0f" % p["value"], {k: p["parity_sample"][k] for k in ("agents","max_rel_dU","frac_dU_le_1e-5")})
for k,v in d["secondary"].items(): print(k, "%.0f solves/s %.1f ms" % (v["value"], v["ms_per_step"]), v.get("converged_frac"), (v.get("parity_sample") or {}).get("max_rel_dU"), (v.get("cpu_baseline") or {}).get("value"))
print("sha", d["controls_sha256_first_65536"][:12], d["secondary"]["pacejka_nx6_N12"]["controls_sha256"][:12])
PY
