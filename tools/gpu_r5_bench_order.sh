#!/bin/bash
# Does the prove_step leg of bench.py depend on what ran before it in the process (the sizes leg allocates and frees ~35 GB)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r5
for round in 1 2 3; do for c in "" "--no-sizes"; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu $c 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['summary']
print('round $round [%-10s]: value %.4f prove %.1f /s (%.4f ms) two chains %.1f bound %.1f compress %.2f' % ('$c' or 'default', d['value'], s['prove_step_per_s'], s['prove_step_ms_median'], s['prove_step_two_chains_per_s'], s['prove_step_bound_form_per_s'], s['compress_ms']))"
done; done 2>&1 | tee gpurun_out/r5/bench_order.txt
