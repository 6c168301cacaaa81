#!/bin/bash
# the whole GPU suite under the switches that select between (bit-identical) code paths
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r03var}; mkdir -p $OUT; cd $R
for V in "MPC_CHAIN_MIN=0" "MPC_NO_CHAIN=1 MPC_LDS_PAIRS=3" "MPC_UNFUSED_EVAL=1 MPC_ALL_ROWS=1" "MPC_NO_MEMO=1 MPC_NO_SPEC=1"; do
  tag=$(echo $V | tr ' =' '__')
  env $V timeout -k 10 600 python -m pytest tests -m gpu -q -x > $OUT/tests_$tag.log 2>&1
  rc=$?
  echo "[$V] rc $rc: $(tail -1 $OUT/tests_$tag.log)"
  if [ $rc -gt 1 ]; then exit 4; fi
done
