#!/bin/bash
# Interleaved A/B of environment tuning switches on ONE box: the MSM leg of bench.py (pipelined and single) and the steady state of
# tools/gpu_prove_time.py, for every configuration given ("" = the defaults), `rounds` times.
# usage (GPU box, repo root): bash tools/ab_env.sh <rounds> "VAR=val ..." "VAR=val ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for round in $(seq 1 $rounds); do
  for cfg in "" "$@"; do
    m=$(env $cfg timeout -k 10 200 python3 $R/bench.py --no-prove --no-cpu --no-sizes --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('msm %.4f GPoints/s  acc alone %.4f ms  single %.4f GPoints/s' % (d['value'] or 0, d['roofline']['avg_launch_ms'], d['summary']['msm_single_gpoints_per_s']))")
    p=$(env $cfg timeout -k 10 300 python3 $R/tools/gpu_prove_time.py 16 100 ref 2>&1 | grep "steady state")
    echo "== round $round [$cfg] $m | prove: $p"
  done
done
