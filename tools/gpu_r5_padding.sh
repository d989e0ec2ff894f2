#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
for cfg in "0 0" "0 1" "0 2" "0 3" "1 1" "2 2" "3 0"; do
  timeout -k 10 120 python3 tools/gpu_single_chain_vs_padding.py $cfg 2>&1 | grep -v amdgpu.ids
done | tee $OUT/single_chain_vs_padding_family.txt
timeout -k 10 900 python -m pytest tests/test_gpu_queues.py -x -q -s > $OUT/pytest_queues.txt 2>&1; rc=$?; echo "pytest rc=$rc"; grep -a "single chain [0-9]\|passed\|failed" $OUT/pytest_queues.txt | tail -4
for cfg in "0 0" "1 1" "3 1"; do
  timeout -k 10 200 python3 tools/gpu_two_chain_conditions.py $cfg 2>&1 | grep -v amdgpu.ids
done | tee $OUT/two_chain_conditions_family.txt
