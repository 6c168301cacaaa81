#!/bin/bash
# the 524 288-agent test, then Pacejka (nx = 6, N = 12, 65 536 agents) with 4 (default) / 5 / 6 / 8 sub-batch groups, alternating
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-r04g}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config4_whole_batch" > $OUT/big.log 2>&1 || { tail -20 $OUT/big.log; exit 3; }
tail -2 $OUT/big.log
STEPS=4 BENCH_ARGS="--model 1 --horizon 12 --no-kernel-pass" bash tools/ab.sh $TAG "" "MPC_GROUPS=6" "MPC_GROUPS=8" "" "MPC_GROUPS=5" "MPC_GROUPS=6" "MPC_GROUPS=8"
