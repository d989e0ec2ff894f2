R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ctx3; mkdir -p $OUT
cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_nova.py tests/test_gpu_field_vec.py -x -q > $OUT/tests.log 2>&1 || { tail -n 30 $OUT/tests.log; exit 1; }
tail -n 2 $OUT/tests.log
for m in 2 1 2 1; do
VDF_NOVA_T_AHEAD=$m python tools/gpu_prove_time.py 16 14 > $OUT/prove_m$m.log 2>&1 || { tail -n 20 $OUT/prove_m$m.log; exit 1; }
echo "mode $m"; tail -n 3 $OUT/prove_m$m.log | head -n 2
done
