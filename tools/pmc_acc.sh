cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_ANY -d $R/gpurun_out/pmc3 -o p -- python $R/tools/gpu_msm_time.py 20 tbl16x1 > $R/gpurun_out/pmc3.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $R/gpurun_out/pmc4 -o p -- python $R/tools/gpu_msm_time.py 20 tbl16x1 > $R/gpurun_out/pmc4.log 2>&1 || exit 1
