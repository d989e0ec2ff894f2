#!/bin/bash
# the GPU suite and the driver's bench command, outputs kept
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_stderr.txt; echo "bench rc=$?"
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
wc -c $OUT/bench_line.json; cat $OUT/bench_line.json
