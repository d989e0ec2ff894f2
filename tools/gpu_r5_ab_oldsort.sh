#!/bin/bash
# prove_step with the library of commit 1e2e107 (before the staged sort; VDF_HIP_LIB) against the shipped one, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
AB=$R/vdf_amd/csrc/build/ab/libvdf_hip.so
for round in 1 2 3 4; do
  for which in old shipped; do
    if [ $which = old ]; then export VDF_HIP_LIB=$AB; else unset VDF_HIP_LIB; fi
    p=$(timeout -k 10 300 python3 tools/gpu_prove_time.py 16 100 ref 2>&1 | grep "steady state")
    echo "== round $round [$which] $p"
  done
done | tee $OUT/ab_old_sort_vs_staged_in_step.txt
