#!/bin/bash
# SQ counters of the Pacejka N = 12 solve on ONE stream (MPC_GROUPS=1: no overlap between kernels):  bash tools/sq_pacejka.sh <tag>
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-sqpac}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MPC_GROUPS=1
X="--no-cpu-baseline --no-kernel-pass --no-pipeline-pass --no-parity-leg --no-secondary --model 1 --horizon 12"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 1 --warmup 0 $X > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_groups1 -- python3 $R/bench.py --steps 2 --warmup 1 $X > $OUT/bench_groups1.json 2> $OUT/stats.err
cd $R
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
o = sys.argv[1]
sq = collections.defaultdict(collections.Counter); nd = collections.Counter()
for f in glob.glob(os.path.join(o, "sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "mpc" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void mpc::", "").replace("mpc::", "")
        sq[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": nd[k] += 1
for k, v in sorted(sq.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    wc = v["SQ_WAVE_CYCLES"]
    if v["SQ_INSTS_VALU"] < 1e6: continue
    print("%-40s dispatches %5d waves %.3g VALU %.4g (per wave %.0f) SALU %.3g VMEM %.3g | wait %.0f %% issue-stall %.0f %% active %.0f %%"
          % (k[:40], nd[k], v["SQ_WAVES"], v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU"] / max(1, v["SQ_WAVES"]), v["SQ_INSTS_SALU"], v["SQ_INSTS_VMEM"],
             100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_ANY"] / wc))
for f in glob.glob(os.path.join(o, "stats_groups1", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:7]:
        print("   %-60s calls %6s total %8.1f ms avg %8.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
