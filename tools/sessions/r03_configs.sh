#!/bin/bash
# round-3 numbers of the other BASELINE configurations and batch sizes (development scripts, one GPU)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r03cfg}; mkdir -p $OUT; cd $R
{
timeout -k 10 200 python tools/dev/config3.py 65536 || exit 3
timeout -k 10 200 python tools/dev/config5.py 32768 || exit 3
X="--no-cpu-baseline --no-kernel-pass --no-pipeline-pass --no-parity-leg --no-secondary --steps 8 --warmup 2"
for b in 4096 8192 16384 32768; do
  timeout -k 10 200 python bench.py $X --batch $b > $OUT/b$b.json 2> $OUT/b$b.err || exit 3
  python - $OUT/b$b.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("batch %d: %.2f ms per solve = %.0f solves/s, groups %s, rounds %s, solo agents %s"%(d["config"]["batch_per_gpu"], d["ms_per_step"], d["value"], d["config"]["sub_batch_groups"], d["solver"]["rounds"], d["solver"]["solo_agents"]))
PY
done
timeout -k 10 200 python tools/dev/closed_loop.py || exit 3
} 2>&1 | tee $OUT/configs.txt
