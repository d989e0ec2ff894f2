#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_msm.py -x -q > $OUT/pytest_sort.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_sort.txt
[ $rc = 0 ] || exit 1
for st in 0 1; do
  echo "== VDF_MSM_SORT_STAGED=$st"
  VDF_MSM_SORT_STAGED=$st timeout -k 10 300 python3 tools/gpu_msm_window_sweep.py "18,20,22,24" "16,17,20" 3 2>&1 | grep -v amdgpu.ids
done | tee $OUT/sort_staged_sweep.txt
