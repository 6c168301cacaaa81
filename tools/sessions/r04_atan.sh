#!/bin/bash
# after the division-free m_atan: parity tests of the model layer + bit-identity tests, then Pacejka and kinematic timings
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-r04at}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_math or rhs or rollout or quad or cost_and_gradient or lost_pacejka or persistent or lookahead or golden" > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 3; }
tail -2 $OUT/tests.log
STEPS=6 BENCH_ARGS="--model 1 --horizon 12" bash tools/ab.sh $TAG "" ""
STEPS=8 bash tools/ab.sh ${TAG}k ""
