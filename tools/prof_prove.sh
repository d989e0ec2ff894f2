cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_prove -o p -- python $GRAFT_REPO_ROOT/tools/gpu_prove_time.py 16 ${1:-10} > $GRAFT_REPO_ROOT/gpurun_out/prof_prove.log 2>&1
