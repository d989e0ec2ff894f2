R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bw; mkdir -p $OUT
cd $R
for c in 16 17 18 15 16; do
VDF_NOVA_BIG_WINDOW=$c python tools/gpu_prove_time.py 16 26 > $OUT/prove_c$c.log 2>&1 || { tail -n 20 $OUT/prove_c$c.log; exit 1; }
echo "primary window $c"; tail -n 2 $OUT/prove_c$c.log | head -n 1
done
