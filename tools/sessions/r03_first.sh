#!/bin/bash
# first GPU session of round 3: tolerance sweep, identical-path statistics, GPU suite, bench
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03a}
mkdir -p $OUT
cd $R
timeout -k 10 300 python tools/dev/eps_sweep.py > $OUT/eps_sweep.txt 2> $OUT/eps_sweep.err || { echo "sweep failed $?"; tail -5 $OUT/eps_sweep.err; exit 3; }
echo sweep done; cat $OUT/eps_sweep.txt
timeout -k 10 200 python tools/dev/paths.py > $OUT/paths.txt 2> $OUT/paths.err || { echo "paths failed $?"; tail -5 $OUT/paths.err; exit 3; }
echo paths done; cat $OUT/paths.txt
timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
rc=$?
tail -40 $OUT/tests.log
if [ $rc -gt 1 ]; then echo "pytest rc $rc"; exit 4; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed $?"; tail -20 $OUT/bench.err; exit 5; }
echo bench done
python - <<'PY'
import json,sys,os
d=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out",sys.argv[1] if len(sys.argv)>1 else "r03a","bench.json")))
print({k:d[k] for k in ("value","ms_per_step")}, d["config"].get("sub_batch_groups"), d["config"].get("streams_side_by_side_measured"))
print("parity_at", {k:v for k,v in d.get("parity_at_1e-5",{}).items() if k!="parity_sample"})
print("secondary", d.get("secondary"))
print("fp64", d["roofline"]["fp64_valu"])
print("pipe", d.get("pipelined_two_handles",{}).get("value"))
PY
