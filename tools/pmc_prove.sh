# VALU instructions per kernel over a short prove loop (one --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_prove -o p -- python $GRAFT_REPO_ROOT/tools/gpu_prove_time.py 16 8 > $GRAFT_REPO_ROOT/gpurun_out/pmc_prove.log 2>&1
