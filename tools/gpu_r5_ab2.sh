#!/bin/bash
# Round 5: in-process interleaved A/B of the fused fold and the lane-serial fix-up; bench after the window change.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
timeout -k 10 500 python3 tools/gpu_prove_ab_inproc.py 7 40 base unfused:fold_fused=0 serial:fixup_serial=1 unfused_serial:fold_fused=0,fixup_serial=1 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_inproc_fold_fixup.txt
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-prove --no-cpu > $OUT/bench_msm_only.json 2> $OUT/bench_msm_only_stderr.txt; echo "bench rc=$?"
cat $OUT/bench_msm_only.json
