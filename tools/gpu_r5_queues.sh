#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_queues.py tests/test_gpu_tuning.py tests/test_gpu_compress.py -x -q -s > $OUT/pytest_queues.txt 2>&1; rc=$?; echo "pytest rc=$rc"; grep -a "single chain\|passed\|failed\|Error" $OUT/pytest_queues.txt | tail -8
[ $rc = 0 ] || { tail -40 $OUT/pytest_queues.txt; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-prove --no-cpu > $OUT/bench_msm_only.json 2> $OUT/bench_msm_only_stderr.txt; echo "bench rc=$?"
cat $OUT/bench_msm_only.json
