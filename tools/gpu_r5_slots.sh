#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
for cfg in "1 2" "2 1" "1 3" "3 1" "2 3" "3 2" "1 4" "4 1" "2 4" "4 2" "3 4" "4 3" "1 5" "5 1"; do
  set -- $cfg
  echo -n "side=$1 critical=$2: "
  VDF_Q_SIDE=$1 VDF_Q_CRITICAL=$2 timeout -k 10 120 python3 tools/gpu_single_chain_vs_padding.py 0 0 2>&1 | grep -v amdgpu.ids
done | tee $OUT/family_slots.txt
