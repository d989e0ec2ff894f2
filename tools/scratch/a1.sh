R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/a1; mkdir -p $OUT
cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_nova.py tests/test_gpu_compress.py tests/test_gpu_seam.py tests/test_gpu_wire.py -x -q > $OUT/tests.log 2>&1 || { tail -n 30 $OUT/tests.log; exit 1; }
tail -n 2 $OUT/tests.log
for m in 1 0 1 0; do
VDF_NOVA_NIFS_AHEAD=$m python tools/gpu_prove_time.py 16 14 > $OUT/prove_m$m.log 2>&1 || { tail -n 20 $OUT/prove_m$m.log; exit 1; }
echo "nifs ahead $m"; tail -n 3 $OUT/prove_m$m.log | head -n 2
done
timeout -k 10 300 python tools/gpu_nova_fuzz.py 20 9 2>&1 | tail -n 2
