#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
for q in 16 32 8; do
for cfg in "0 0" "1 1" "3 1"; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 tools/gpu_two_chain_conditions.py $cfg 2>&1 | grep -v amdgpu.ids
done; done | tee $OUT/two_chain_conditions_2.txt
