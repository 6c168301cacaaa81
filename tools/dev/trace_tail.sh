#!/bin/bash
# kernel trace of bench.py's solve: when do the persistent-kernel launches start and end relative to the rounds?
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-tt}; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
X="--no-cpu-baseline --no-kernel-pass --no-pipeline-pass --no-parity-leg --no-secondary"
i=0
for V in "$@"; do
  i=$((i+1)); rm -rf $OUT/tr$i
  env $V rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$i -- python3 $R/bench.py --steps 1 --warmup 1 $X > $OUT/b$i.json 2> $OUT/e$i.err
  python3 - "$V" $OUT/tr$i <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "mpc" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mpc::", "").replace("mpc::", "").split("<")[0], int(r.get("Grid_Size", 0) or 0)))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "init_kernel"]
rows = rows[starts[-1]:]
t0 = rows[0][0]
last_step = max(r[1] for r in rows if r[2] == "step_kernel")
end = max(r[1] for r in rows)
print("[%s] solve %.1f ms; last step kernel ends at %.1f ms" % (sys.argv[1], (end - t0) / 1e6, (last_step - t0) / 1e6))
for r in rows:
    if r[2] in ("solo_kernel", "solo_select_kernel"):
        print("    %-20s grid %6d  start %7.2f ms  end %7.2f ms  (%.2f ms)" % (r[2], r[3], (r[0] - t0) / 1e6, (r[1] - t0) / 1e6, (r[1] - r[0]) / 1e6))
PY
done
