# kernel timeline of steady-state prove_steps at t = 2^16 (start, end, duration in us, stream): bash tools/gpu_step_timeline.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/p
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/p -o p -- python $R/tools/gpu_prove_time.py 16 10 > $OUT/prove_prof.log 2>&1 || exit 1
python $R/tools/timeline.py $(ls $OUT/p/*.db | head -1) k_nifs_cross 10 6 > $OUT/timeline.txt 2>&1
rm -rf $OUT/p
grep -E "k_nifs_cross|k_spmv_long" $OUT/timeline.txt
