#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_tuning.py tests/test_gpu_compress.py -x -q > gpurun_out/r5/pytest_fill3.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5/pytest_fill3.txt
for round in 1 2; do for e in "VDF_MSM_ACC_WG=2" "VDF_MSM_ACC_WG=0"; do
  env $e timeout -k 10 300 python3 bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['summary']
print('round $round $e: value %.4f acc alone %.4f ms single %.4f | vb %.4f | 2^20 %.4f 2^22 %.4f (2x %.4f) 2^24 %.4f (2x %.4f) | prove %.1f two %.1f compress %.2f' % (d['value'], d['roofline']['avg_launch_ms'], s['msm_single_gpoints_per_s'], s['msm_variable_base_2_20_gpoints_per_s'], s['msm_table_2_20_gpoints_per_s'], s['msm_table_2_22_gpoints_per_s'], s['msm_table_2_22_two_in_flight_gpoints_per_s'], s['msm_table_2_24_gpoints_per_s'], s['msm_table_2_24_two_in_flight_gpoints_per_s'], s['prove_step_per_s'], s['prove_step_two_chains_per_s'], s['compress_ms']))"
done; done 2>&1 | tee gpurun_out/r5/fill_auto_ab.txt
