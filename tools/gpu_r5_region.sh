#!/bin/bash
# The MSM leg's value by region length and steps in flight (one box): what a region's start and end cost.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r5
for d in 3 2 4; do for k in 20 40 100; do
  timeout -k 10 200 python3 bench.py --no-prove --no-cpu --no-sizes --depth $d --steps $k --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('depth $d steps $k: value %.4f  ms/step %.4f  region %.3f ms' % (d['value'], d['ms_per_step'], d['ms_per_step']*$k))"
done; done 2>&1 | tee gpurun_out/r5/region_sweep.txt
