#!/bin/bash
# effective shader clock per kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS give-back)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-clock}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MPC_GROUPS=1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-pass > $OUT/bench_grbm.json 2> $OUT/grbm.err
cd $R
python3 - <<PY
import csv, glob, collections
dur = {}
for f in glob.glob("$OUT/grbm/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob("$OUT/grbm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "mpc" not in r["Kernel_Name"]: continue
        d = dur.get(r["Dispatch_Id"])
        if not d or d[0] < 20000: continue   # the quotient reads high on short dispatches
        k = r["Kernel_Name"].split("(")[0].replace("void mpc::", "").split("<")[0]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += d[0]; acc[k][2] += 1
for k, (cyc, ns, n) in acc.items():
    print("%-22s dispatches %5d  mean duration %7.1f us  effective clock %.2f GHz" % (k, n, ns / n / 1e3, cyc / 8 / ns))
PY
