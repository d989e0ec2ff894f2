R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/mode2; mkdir -p $OUT
cd $R
for m in 1 2 1 2; do
VDF_NOVA_T_AHEAD=$m python tools/gpu_prove_time.py 16 14 > $OUT/prove_m$m.log 2>&1 || { tail -n 20 $OUT/prove_m$m.log; exit 1; }
echo "mode $m"; tail -n 3 $OUT/prove_m$m.log | head -n 2
done
