#!/bin/bash
# A/B timing on the GPU box:  bash tools/ab.sh <tag> "ENV1=.. ENV2=.." "ENV=.." ...   (one bench.py run per argument; "" = defaults)
# prints value / ms_per_step / per-kernel ms of the single-group pass / checksum per variant
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
i=0
for V in "$@"; do
  i=$((i+1))
  env $V timeout -k 10 200 python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-pipeline-pass --no-parity-leg --no-secondary ${BENCH_ARGS} > $OUT/ab_$i.json 2> $OUT/ab_$i.err || { echo "variant '$V' failed"; tail -5 $OUT/ab_$i.err; exit 3; }
  python - "$V" $OUT/ab_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
k=d.get("kernels",{})
print("[%s] %.0f solves/s %.2f ms | one-stream ms:"%(sys.argv[1],d["value"],d["ms_per_step"]), {n.replace("_kernel",""):round(v["ms_per_solve"],1) for n,v in k.items()}, "rounds",d["solver"]["rounds"],"sha",d["controls_sha256_first_65536"][:12], flush=True)
PY
done
