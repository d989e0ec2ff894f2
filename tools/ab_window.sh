#!/bin/bash
# Window sweep of the fixed-base MSM of bench.py at 2^k points on ONE box (pipelined and one-at-a-time), `rounds` times.
# usage (GPU box, repo root): bash tools/ab_window.sh <rounds> <log2n> <window> <window> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; lg=$2; shift 2
for round in $(seq 1 $rounds); do
  for w in "$@"; do
    m=$(timeout -k 10 200 python3 $R/bench.py --no-prove --no-cpu --no-sizes --steps 60 --log2n $lg --window $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('msm %.4f GPoints/s  acc alone %.4f ms  single %.4f GPoints/s' % (d['value'] or 0, d['roofline']['avg_launch_ms'], d['summary']['msm_single_gpoints_per_s']))")
    echo "== round $round 2^$lg window $w: $m"
  done
done
