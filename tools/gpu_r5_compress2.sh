#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_wire.py tests/test_gpu_snark.py -x -q > $OUT/pytest_compress.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_compress.txt
[ $rc = 0 ] || exit 1
timeout -k 10 120 python3 tools/gpu_compress_time.py 16 > $OUT/compress_per_kernel_hosttail.txt 2>&1; head -22 $OUT/compress_per_kernel_hosttail.txt
