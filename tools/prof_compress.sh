cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_compress -o p -- python $GRAFT_REPO_ROOT/tools/gpu_compress_time.py 16 > $GRAFT_REPO_ROOT/gpurun_out/prof_compress.log 2>&1
