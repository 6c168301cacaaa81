#!/bin/bash
# K1 alone at growing batch sizes under rocprofv3 --pmc (kernel-trace only): per-launch SQ / instruction-cache counters
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; OUT=$R/gpurun_out/${1:-k1scan}; mkdir -p $OUT
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_BUSY_CU_CYCLES" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1)); rm -rf $OUT/pmc$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $R/tools/dev/k1_scan.py ${SIZES} > $OUT/pmc$i.log 2> $OUT/pmc$i.err || { tail -5 $OUT/pmc$i.err; exit 3; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
val = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mpc::", "")
        if n.startswith(("rollout", "stage_adjoint")):
            val[(n[:24], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(val):
    d = {c: sorted(v)[len(v) // 2] for c, v in val[k].items()}
    print(k, " ".join("%s=%.4g" % (c.replace("SQ_", "").replace("SQC_", "C_"), d[c]) for c in sorted(d)))
PY
