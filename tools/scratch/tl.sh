R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/p
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/p -o p -- python $R/tools/gpu_prove_time.py 16 10 > $OUT/prove_prof.log 2>&1 || exit 1
python $R/tools/timeline.py $(ls $OUT/p/*.db | head -1) k_nifs_cross 10 4 > $OUT/timeline.txt 2>&1
rm -rf $OUT/p
cat $OUT/timeline.txt
