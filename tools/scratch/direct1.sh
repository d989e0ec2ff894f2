R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/direct; mkdir -p $OUT
cd $R && timeout -k 10 600 python -m pytest tests/test_gpu_msm.py -x -q -k "digit_table" > $OUT/tests.log 2>&1; tail -n 25 $OUT/tests.log
