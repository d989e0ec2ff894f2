#!/bin/bash
# Round 5: lane-serial fix-up (VDF_MSM_FIXUP_SERIAL) and the sort geometry of windows >= 18: parity, sweep, A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
VDF_MSM_FIXUP_SERIAL=1 timeout -k 10 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_nova.py -x -q > $OUT/pytest_fixup_serial.txt 2>&1; rc=$?; echo "pytest (serial fix-up) rc=$rc"; tail -3 $OUT/pytest_fixup_serial.txt
[ $rc = 0 ] || exit 1
timeout -k 10 600 python3 tools/gpu_msm_window_sweep.py "20,21,22,24" "17,18,19,20,21,22" 3 2>&1 | grep -v amdgpu.ids | tee $OUT/window_sweep_after.txt
VDF_MSM_FIXUP_SERIAL=1 timeout -k 10 300 python3 tools/gpu_msm_window_sweep.py "20,22" "17,20" 3 2>&1 | grep -v amdgpu.ids | tee $OUT/window_sweep_serial.txt
bash tools/ab_env.sh 2 "VDF_MSM_FIXUP_SERIAL=1" 2>&1 | tee $OUT/ab_fixup_serial.txt
