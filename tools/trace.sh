#!/bin/bash
# kernel trace of one solve (G=1): per-kernel duration by round, to see where a round's time goes
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MPC_GROUPS=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/dev/perf.py 0 20 ${TRACE_B:-65536} 0 1 1 > $OUT/run.log 2> $OUT/run.err
cd $R
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mpc" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mpc::", "").split("<")[0]))
rows.sort()
by = collections.defaultdict(list)
for s, e, k in rows: by[k].append((e - s) / 1e3)
for k, v in by.items():
    n = len(v)
    if n < 10: continue
    q = lambda p: sorted(v)[int(p * (n - 1))]
    print("%-16s n=%d mean %.1f us  p10 %.1f p50 %.1f p90 %.1f max %.1f" % (k, n, sum(v) / n, q(.1), q(.5), q(.9), max(v)))
    # by round decile
    step = n // 10
    print("    by tenth of the solve:", " ".join("%.0f" % (sum(v[i * step:(i + 1) * step]) / step) for i in range(10)))
# gaps between consecutive kernels
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
print("gaps: mean %.2f us, total %.1f ms, span %.1f ms" % (sum(gaps) / len(gaps) / 1e3, sum(gaps) / 1e6, (rows[-1][1] - rows[0][0]) / 1e6))
PY
