#!/bin/bash
# full GPU suite, bench, pipeline A/B, 2-rank same-GPU rehearsal (napping vs spinning round loop), profile recipe
R=$GRAFT_REPO_ROOT
TAG=${1:-r03h}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
rc=$?
tail -5 $OUT/tests.log
if [ $rc -gt 1 ]; then echo "pytest rc $rc"; exit 4; fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -20 $OUT/bench.err; exit 5; }
echo bench done
# the two-handle pipeline with and without the thread-per-agent blocks, napping and spinning host loop
Y="--no-cpu-baseline --no-kernel-pass --no-parity-leg --no-secondary --steps 8 --warmup 2"
for V in "" "MPC_NO_CHAIN=1" "MPC_SPIN=1"; do
  env $V timeout -k 10 200 python bench.py $Y > $OUT/pipe.json 2> $OUT/pipe.err || { echo "pipe failed"; tail -5 $OUT/pipe.err; exit 5; }
  python - "$V" $OUT/pipe.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print("[%s] blocking %.0f  two handles %.0f solves/s"%(sys.argv[1], d["value"], d["pipelined_two_handles"]["value"]), flush=True)
PY
done
X="--no-cpu-baseline --no-kernel-pass --no-pipeline-pass --no-parity-leg --no-secondary"
TIMEFORMAT="%U user %S sys %R wall (seconds; both ranks and their launcher, start-up included)"
for mode in nap spin; do
  if [ $mode = spin ]; then export MPC_SPIN=1; else unset MPC_SPIN; fi
  { time timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --same-gpu --steps 12 --warmup 2 $X > $OUT/rehearsal_$mode.json 2> $OUT/rehearsal_$mode.err ; } 2> $OUT/rehearsal_$mode.time
  if [ ! -s $OUT/rehearsal_$mode.json ]; then echo "rehearsal $mode failed"; tail -20 $OUT/rehearsal_$mode.err; exit 6; fi
  echo "rehearsal $mode: $(cat $OUT/rehearsal_$mode.time)"
done
unset MPC_SPIN
python - $OUT <<'PY'
import json,sys,os
o=sys.argv[1]
d=json.load(open(os.path.join(o,"bench.json")))
print("bench", round(d["value"]), round(d["ms_per_step"],2), "groups", d["config"].get("sub_batch_groups"), "fp64 frac", round(d["roofline"]["flops_fraction"],4))
print("kernels", {k:round(v["ms_per_solve"],1) for k,v in d["kernels"].items()})
print("parity_at", {k:v for k,v in d.get("parity_at_1e-5",{}).items() if k in("alm_eps","value","frac_dU_le_1e-5","ms_per_step")})
s=d.get("secondary",{}).get("pacejka_nx6_N12",{}); print("secondary", {k:s.get(k) for k in("value","ms_per_step","solo_kernel_ms_longest","solo_kernel_share","converged_frac")})
print("pipe", round(d.get("pipelined_two_handles",{}).get("value",0)), "cpu", d.get("cpu_baseline",{}).get("value"))
for m in ("nap","spin"):
    r=[l for l in open(os.path.join(o,"rehearsal_%s.json"%m)) if l.startswith("{")]
    r=json.loads(r[-1]); print("rehearsal",m,round(r["value"]),round(r["ms_per_step"],2),r["controls_sha256_first_65536"][:12])
PY
bash tools/profile.sh $TAG > $OUT/profile.log 2>&1 || { echo "profile failed"; tail -20 $OUT/profile.log; exit 7; }
tail -3 $OUT/profile.log
