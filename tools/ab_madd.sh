#!/bin/bash
# A/B of the bucket loop's mixed addition on ONE box, interleaved: a second build of the library (`make -C vdf_amd/csrc ab
# AB_FLAGS=-DVDF_MADD_V1`: round 3's ten products and seven subtractions) against the shipped one.  The second build is
# SELECTED with VDF_HIP_LIB (vdf_amd/_lib.py); the shipped vdf_amd/libvdf_hip.so is never touched.
# usage (GPU box, repo root): bash tools/ab_madd.sh [rounds]
set -eu
R=${GRAFT_REPO_ROOT:-$(pwd)}
AB=$R/vdf_amd/csrc/build/ab/libvdf_hip.so
[ -f "$AB" ] || { echo "build the A/B library first: make -C vdf_amd/csrc ab"; exit 1; }
for round in $(seq 1 ${1:-2}); do
  for which in ab shipped; do
    if [ $which = ab ]; then export VDF_HIP_LIB=$AB; else unset VDF_HIP_LIB; fi
    m=$(timeout -k 10 200 python3 $R/bench.py --no-prove --no-cpu --no-sizes --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('msm %.4f GPoints/s  %.4f ms/step  acc alone %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))")
    p=$(timeout -k 10 300 python3 $R/tools/gpu_prove_time.py 16 100 ref 2>&1 | grep "steady state")
    echo "== round $round [$which] $m | prove: $p"
  done
done
