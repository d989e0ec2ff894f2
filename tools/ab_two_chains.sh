#!/bin/bash
# Two concurrent chains under the shipped library and under the A/B build (VDF_HIP_LIB), with the digit budgets of interest.
# usage (GPU box, repo root): bash tools/ab_two_chains.sh
set -eu
R=${GRAFT_REPO_ROOT:-$(pwd)}
AB=$R/vdf_amd/csrc/build/ab/libvdf_hip.so
run() { echo "== $1"; shift; env "$@" timeout -k 10 280 python3 $R/tools/gpu_prove_two_chains.py 16 60 2 2>&1 | grep "chain" || true; }
run "shipped, defaults (20 GiB digit budget)" X=1
run "shipped, 72 GiB" VDF_NOVA_DIGIT_BUDGET_GIB=72
if [ -f "$AB" ]; then
  run "A/B build, 72 GiB" VDF_HIP_LIB=$AB VDF_NOVA_DIGIT_BUDGET_GIB=72
fi
