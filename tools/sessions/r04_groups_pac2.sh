#!/bin/bash
# Pacejka 65 536 agents: four (default) vs five sub-batch groups, alternating; then the kinematic batch with five once
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-r04g2}; cd $R
STEPS=6 BENCH_ARGS="--model 1 --horizon 12 --no-kernel-pass" bash tools/ab.sh $TAG "" "MPC_GROUPS=5" "" "MPC_GROUPS=5" "" "MPC_GROUPS=5"
STEPS=8 BENCH_ARGS="--no-kernel-pass" bash tools/ab.sh ${TAG}k "" "MPC_GROUPS=5" "" "MPC_GROUPS=5"
