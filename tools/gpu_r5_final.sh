#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_final.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu_final.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
