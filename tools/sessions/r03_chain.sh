#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03c}
mkdir -p $OUT
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chained or persistent or switch_points or memo or full_size_prop or solve_matches" > $OUT/tests_chain.log 2>&1
rc=$?
tail -30 $OUT/tests_chain.log
if [ $rc -ne 0 ]; then echo "pytest rc $rc"; exit 4; fi
bash tools/ab.sh ${1:-r03c} "" "MPC_NO_CHAIN=1" ""
