#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5; mkdir -p $OUT; cd $R
for round in 1 2; do
for cfg in "X=1" "VDF_MSM_ACC_WG=3" "VDF_MSM_ACC_WG=1"; do
  echo "== [$cfg] $(env $cfg timeout -k 10 120 python3 tools/gpu_compress_time.py 16 2>&1 | grep "^compress" | tr '\n' '|')"
done; done | tee $OUT/compress_fill.txt
