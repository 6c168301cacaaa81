#!/bin/bash
# K1 alone at growing batch sizes under rocprofv3 --kernel-trace: launch time against the number of waves
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; OUT=$R/gpurun_out/${1:-k1scan}; mkdir -p $OUT; rm -rf $OUT/k1scan
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/k1scan -- python3 $R/tools/dev/k1_scan.py > $OUT/k1scan.log 2> $OUT/k1scan.err || { tail -5 $OUT/k1scan.err; exit 3; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/k1scan/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mpc::", "")
    if n.startswith(("rollout", "stage", "adjoint")):
        by[(n, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), d in sorted(by.items()):
    print("%-36s grid %8d  n %2d  min %.1f  median %.1f us" % (n[:36], g, len(d), min(d), sorted(d)[len(d) // 2]))
PY
