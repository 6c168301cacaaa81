#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):   bash tools/profile.sh r02a
#   1. rocprofv3 --kernel-trace --stats of bench.py            -> kernel_stats.csv
#   2. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes     -> pmc_raw.json  (HBM traffic per kernel)
#   3. SQ counters with one sub-batch group (no overlap)       -> sq_counters.txt
#   4. kernel trace of one solve, one group, persistent kernel on / off -> tail.txt
# Outputs land in gpurun_out/<tag>/ ; tools/pmc_summary.py copies the summaries to profiles/<tag>_*.
# (counter passes never combine --pmc with the hip/hsa trace domains: kernel-trace only)
set -e
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-pass --no-pipeline-pass --no-parity-leg --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 $B > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 $B > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 $B > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo write done
export MPC_GROUPS=1
# the same statistics on ONE stream: what bench.py's per-kernel figures (its untimed single-group pass) must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_groups1 -- python3 $R/bench.py --steps 2 --warmup 1 $B > $OUT/bench_groups1_under_rocprof.json 2> $OUT/stats_groups1.err
echo stats groups1 done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 1 --warmup 0 $B > $OUT/bench_sq.json 2> $OUT/sq.err || true
echo sq done
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_solo -- python3 $R/bench.py --steps 1 --warmup 1 $B > $OUT/bench_trace_solo.json 2> $OUT/trace_solo.err
MPC_SOLO_MAX=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_rounds -- python3 $R/bench.py --steps 1 --warmup 1 $B > $OUT/bench_trace_rounds.json 2> $OUT/trace_rounds.err
unset MPC_GROUPS
echo trace done
cd $R
python3 tools/pmc_summary.py $TAG
