R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/early; mkdir -p $OUT
cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_nova.py tests/test_gpu_compress.py tests/test_gpu_seam.py tests/test_gpu_wire.py -x -q > $OUT/tests.log 2>&1 || { tail -n 30 $OUT/tests.log; exit 1; }
tail -n 2 $OUT/tests.log
VDF_NOVA_SYNTH_TRACE=1 python tools/gpu_prove_time.py 16 12 > $OUT/prove_trace.log 2>&1 || { tail -n 20 $OUT/prove_trace.log; exit 1; }
grep "synth side" $OUT/prove_trace.log | tail -n 4; tail -n 3 $OUT/prove_trace.log
python tools/gpu_prove_time.py 16 12 > $OUT/prove.log 2>&1; tail -n 3 $OUT/prove.log
