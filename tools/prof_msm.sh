cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_msm -o p -- python $GRAFT_REPO_ROOT/tools/gpu_msm_time.py ${1:-20} ${2:-tbl16x1} > $GRAFT_REPO_ROOT/gpurun_out/prof_msm.log 2>&1
