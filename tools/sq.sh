#!/bin/bash
# SQ counters of one solve on ONE stream (no overlap) for the current environment switches:
#   bash tools/sq.sh <tag>     -> gpurun_out/<tag>/sq/ ; summarise with  python tools/pmc_summary.py <tag>
set -e
TAG=${1:-sq}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MPC_GROUPS=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-pass --no-pipeline-pass > $OUT/bench_sq.json 2> $OUT/sq.err
echo sq done
