#!/bin/bash
# Development: per-kernel time of K1 alone (tools/dev/k1b.py) for a list of library builds
#   bash tools/k1b.sh base xNOTAN ...     (gpurun_<name>.so in the repo root; "base" = the product library)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export MPC_UNFUSED_EVAL=1
for v in "$@"; do
  if [ $v = base ]; then unset MPC_LIB_PATH; else export MPC_LIB_PATH=$R/gpurun_$v.so; fi
  mkdir -p $R/gpurun_out/k1b/$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k1b/$v -- python3 $R/tools/dev/k1b.py > /dev/null 2> $R/gpurun_out/k1b/$v/err.txt
  echo "== $v"
  python3 - $R/gpurun_out/k1b/$v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mpc::" in r["Name"] and int(r["Calls"]) >= 20:
            print("   %-28s calls %5s avg %8.1f us" % (r["Name"].split("(")[0].replace("void ", "").replace("mpc::", ""), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
