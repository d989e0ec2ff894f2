R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tune; mkdir -p $OUT
cd $R && timeout -k 10 600 python -m pytest tests/test_gpu_msm.py -x -q > $OUT/msm_tests.log 2>&1 || { tail -n 20 $OUT/msm_tests.log; exit 1; }
tail -n 2 $OUT/msm_tests.log
cd /tmp && export TMPDIR=/tmp
for cfg in new old; do
  if [ $cfg = old ]; then export VDF_MSM_HEAVY_MIN=24 VDF_MSM_GIANT_SPAN=1024; fi
  rm -rf $OUT/p_$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/p_$cfg -o p -- python $R/tools/gpu_prove_time.py 16 10 > $OUT/prove_$cfg.log 2>&1 || exit 1
  python $R/tools/timeline.py $(ls $OUT/p_$cfg/*.db | head -1) k_nifs_cross 6 > $OUT/timeline_$cfg.txt 2>&1
  tail -n 2 $OUT/prove_$cfg.log
  rm -rf $OUT/p_$cfg
done
unset VDF_MSM_HEAVY_MIN VDF_MSM_GIANT_SPAN
cd $R && timeout -k 10 300 python bench.py --no-prove --no-cpu 2>/dev/null | tail -c 1500
