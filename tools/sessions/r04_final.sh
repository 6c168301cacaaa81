#!/bin/bash
# round 4 closing session: suite, bench line (20 steps), 2-rank rehearsal, the profiling recipe on the final build
R=$GRAFT_REPO_ROOT
TAG=${1:-r04p}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
echo "suite rc $?"; tail -4 $OUT/tests.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -20 $OUT/bench.err; exit 5; }
echo bench done
X="--no-cpu-baseline --no-kernel-pass --no-pipeline-pass --no-parity-leg --no-secondary"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --same-gpu --steps 12 --warmup 2 $X > $OUT/rehearsal.json 2> $OUT/rehearsal.err || { echo "rehearsal failed"; tail -20 $OUT/rehearsal.err; }
echo rehearsal done
bash tools/profile.sh $TAG > $OUT/profile.log 2>&1 || { echo "profile failed"; tail -20 $OUT/profile.log; }
echo profile done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pacejka -- python3 $R/bench.py --model 1 --horizon 12 --steps 2 --warmup 1 $X > $OUT/bench_pacejka_under_rocprof.json 2> $OUT/stats_pacejka.err
echo pacejka stats done
cd $R
python - $OUT <<'PY'
import json,sys,os
o=sys.argv[1]
d=json.load(open(os.path.join(o,"bench.json")))
print("bench", round(d["value"]), round(d["ms_per_step"],2), "groups", d["config"].get("sub_batch_groups"), "fp64 useful", round(d["roofline"]["flops_fraction"],4), "traffic", d["roofline"]["traffic"])
print("kernels", {k:round(v["ms_per_solve"],1) for k,v in d["kernels"].items()})
p=d["parity_at_1e-5"]; print("parity leg %.0f" % p["value"], {k: p["parity_sample"][k] for k in ("agents","max_rel_dU","frac_dU_le_1e-5")})
for k,v in d["secondary"].items(): print(k, "%.0f solves/s %.1f ms" % (v["value"], v["ms_per_step"]), v.get("converged_frac"), v.get("solo_kernel_share"))
print("pipe", round(d.get("pipelined_two_handles",{}).get("value",0)), "cpu", d.get("cpu_baseline",{}).get("value"))
r=[l for l in open(os.path.join(o,"rehearsal.json")) if l.startswith("{")]
if r:
    r=json.loads(r[-1]); print("rehearsal",round(r["value"]),round(r["ms_per_step"],2),r["controls_sha256_first_65536"][:12])
PY
