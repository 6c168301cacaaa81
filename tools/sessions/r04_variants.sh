#!/bin/bash
# the whole GPU suite under the switches that select between bit-identical code paths (final build of round 4)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r04v}
mkdir -p $OUT
cd $R
i=0
for V in "MPC_NO_LOOKAHEAD=1 MPC_CHAIN_MIN=0" "MPC_NO_CHAIN=1 MPC_LDS_PAIRS=3" "MPC_UNFUSED_EVAL=1 MPC_ALL_ROWS=1" "MPC_NO_MEMO=1 MPC_NO_SPEC=1"; do
  i=$((i+1))
  env $V timeout -k 10 600 python -m pytest tests -m gpu -q -x > $OUT/tests_$i.log 2>&1
  echo "[$V] rc $? $(tail -1 $OUT/tests_$i.log)"
done
