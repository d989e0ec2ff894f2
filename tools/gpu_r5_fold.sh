#!/bin/bash
# Round 5: the fused fold (stencil with the previous fold on the way): parity tests, then A/B of the step rate and the time lines.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_field_vec.py tests/test_gpu_nova.py tests/test_gpu_tuning.py tests/test_gpu_c_client.py tests/test_gpu_msm.py -x -q > $OUT/pytest_fold.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest_fold.txt
[ $rc = 0 ] || exit 1
for round in 1 2 3; do
  for cfg in "VDF_NOVA_FOLD_FUSED=0" "VDF_NOVA_FOLD_FUSED=1"; do
    p=$(env $cfg timeout -k 10 300 python3 tools/gpu_prove_time.py 16 100 ref 2>&1 | grep "steady state")
    echo "== round $round [$cfg] prove: $p"
  done
done | tee $OUT/ab_fold_fused.txt
VDF_NOVA_FOLD_FUSED=0 timeout -k 10 200 python3 tools/gpu_step_events.py 16 ref > $OUT/events_fold_unfused.txt 2>&1
VDF_NOVA_FOLD_FUSED=1 timeout -k 10 200 python3 tools/gpu_step_events.py 16 ref > $OUT/events_fold_fused.txt 2>&1
tail -5 $OUT/events_fold_fused.txt
