R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/direct; mkdir -p $OUT
cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_nova.py -x -q -k "digit_table or nova" > $OUT/tests3.log 2>&1 || { tail -n 30 $OUT/tests3.log; exit 1; }
tail -n 2 $OUT/tests3.log
for c in 10 11; do
VDF_NOVA_DIGIT_WINDOW=$c python tools/gpu_prove_time.py 16 12 > $OUT/prove_c$c.log 2>&1 || { tail -n 20 $OUT/prove_c$c.log; exit 1; }
echo "digit window $c"; head -n 1 $OUT/prove_c$c.log | cut -c1-40; tail -n 3 $OUT/prove_c$c.log
done
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/p
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/p -o p -- python $R/tools/gpu_prove_time.py 16 10 > $OUT/prove_prof.log 2>&1 || exit 1
python $R/tools/timeline.py $(ls $OUT/p/*.db | head -1) k_nifs_cross 9 > $OUT/timeline.txt 2>&1
rm -rf $OUT/p
cat $OUT/timeline.txt
