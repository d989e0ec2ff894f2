#!/bin/bash
# A/B of prove_step tuning switches on ONE box (boxes differ by +-5 %): steady-state lines only.  usage: gpu_prove_ab.sh "VAR=val ..." ...
mkdir -p gpurun_out/r3
for cfg in "" "$@"; do
  for rep in 1 2; do
    echo "== [$cfg] run $rep: $(env $cfg python tools/gpu_prove_time.py 16 100 ref 2>&1 | grep 'steady state')"
  done
done
