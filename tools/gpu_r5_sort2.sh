#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_msm.py -x -q > $OUT/pytest_sort.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_sort.txt
[ $rc = 0 ] || exit 1
timeout -k 10 300 python3 tools/gpu_msm_window_sweep.py "16,18,20,22,24" "16,17,20" 3 2>&1 | grep -v amdgpu.ids | tee $OUT/sort_staged_sweep2.txt
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-prove --no-cpu > $OUT/bench_msm_only.json 2> $OUT/bench_msm_only_stderr.txt; echo "bench rc=$?"
python3 -c "import json; d=json.load(open('$OUT/bench_msm_only.json')); print(d['value'], d['msm_sizes'])"
