#!/bin/bash
R=$GRAFT_REPO_ROOT
TAG=${1:-r04c}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "iterate_prefix or path_agreement or in_place_centerline or wall_clock or getters_are_refused" > $OUT/tests_new.log 2>&1
rc=$?; grep -n "iterate-prefix parity\|identical paths" $OUT/tests_new.log | cut -c1-1500; tail -5 $OUT/tests_new.log; echo "new tests rc $rc"
timeout -k 10 500 python tools/dev/first_divergence.py --out $OUT/first_divergence.txt > $OUT/first_divergence.log 2> $OUT/first_divergence.err || { echo "first_divergence failed"; tail -20 $OUT/first_divergence.err; }
grep -v "^     agent" $OUT/first_divergence.txt
