cd /tmp && export TMPDIR=/tmp
for c in 18 19 20; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c$c -o p -- python $GRAFT_REPO_ROOT/tools/gpu_msm_time.py 20 tbl${c}x1 > $GRAFT_REPO_ROOT/gpurun_out/prof_c$c.log 2>&1 || exit 1
done
