#!/bin/bash
# Development: per-kernel time of K1 alone at one request count per run:  bash tools/k1a.sh 4096 16384 ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export MPC_UNFUSED_EVAL=1 MPC_WIDE_MAX=-1
for B in "$@"; do
  mkdir -p $R/gpurun_out/k1a/$B
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k1a/$B -- python3 $R/tools/dev/k1a_scaling.py $B > /dev/null 2> $R/gpurun_out/k1a/$B/err.txt
  python3 - $R/gpurun_out/k1a/$B $B <<'PY'
import csv, glob, sys
out = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mpc::" in r["Name"] and int(r["Calls"]) >= 10 and "cl_" not in r["Name"]:
            out.append("%s %.1f" % (r["Name"].split("(")[0].replace("void ", "").replace("mpc::", "")[:20], float(r["AverageNs"]) / 1e3))
print("requests", sys.argv[2], " | ".join(sorted(out)))
PY
done
