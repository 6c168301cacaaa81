#!/bin/bash
# after the division-free kinematic stage inputs: the model-layer parity tests, the bit-identity tests between the rollout
# variants, the solve parity tests; then 65 536 agents against the control library (HEAD), alternating
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-r04pr}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size and not config3 and not config4 and not config5" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 3; }
tail -2 $OUT/tests.log
D="MPC_LIB_PATH=$R/gpurun_libmpc_ctl.so"
STEPS=8 bash tools/ab.sh $TAG "" "$D" "" "$D" "" "$D"
