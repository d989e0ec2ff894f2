#!/bin/bash
# per-kernel times of one MSM at 2^22 under windows 17..20 (rocprofv3 --kernel-trace, rocpd database -> tools/kstats.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in ${1:-17 18 19 20}; do
  rm -rf /tmp/prof_c$c
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/prof_c$c -o p -- python3 $R/tools/gpu_msm_window_sweep.py ${2:-22} $c 3 > $OUT/sortprof_c$c.log 2>&1
  db=$(find /tmp/prof_c$c -name "*.db" | head -1)
  echo "== window $c ($(grep window $OUT/sortprof_c$c.log | tail -1))"
  python3 $R/tools/kstats.py $db | grep -v "generate\|precompute\|Cijk\|elementwise\|distribution\|index" | head -16
done
