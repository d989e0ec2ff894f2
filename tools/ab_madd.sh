#!/bin/bash
# A/B of the bucket loop's mixed addition on ONE box, interleaved: ab/libvdf_hip_v1.so (built with -DVDF_MADD_V1: round 3's ten
# products and seven subtractions) against the shipped library (product pair for Y3, sign-tracked accumulator).
# usage (GPU box, repo root): bash tools/ab_madd.sh [rounds]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4
mkdir -p $OUT
cp $R/vdf_amd/libvdf_hip.so /tmp/libvdf_hip_new.so || exit 1
for round in $(seq 1 ${1:-2}); do
  for which in v1 new; do
    if [ $which = v1 ]; then cp $R/ab/libvdf_hip_v1.so $R/vdf_amd/libvdf_hip.so; else cp /tmp/libvdf_hip_new.so $R/vdf_amd/libvdf_hip.so; fi
    m=$(timeout -k 10 200 python3 $R/bench.py --no-prove --no-cpu --no-sizes --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('msm %.4f GPoints/s  %.4f ms/step  acc alone %.4f ms  single %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['single_msm']['latency_ms']))")
    p=$(timeout -k 10 300 python3 $R/tools/gpu_prove_time.py 16 100 ref 2>&1 | grep "steady state")
    echo "== round $round [$which] $m | prove: $p"
  done
done
cp /tmp/libvdf_hip_new.so $R/vdf_amd/libvdf_hip.so
