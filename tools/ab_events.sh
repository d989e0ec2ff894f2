#!/bin/bash
# Every launch of two steps on one time line, for the A/B build (VDF_HIP_LIB) and for the shipped library.
# usage (GPU box, repo root): bash tools/ab_events.sh [outdir]
set -eu
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/r5}
AB=$R/vdf_amd/csrc/build/ab/libvdf_hip.so
[ -f "$AB" ] || { echo "build the A/B library first: make -C vdf_amd/csrc ab"; exit 1; }
mkdir -p $OUT
VDF_HIP_LIB=$AB timeout -k 10 200 python3 $R/tools/gpu_step_events.py 16 ref > $OUT/events_ab.txt 2>&1
timeout -k 10 200 python3 $R/tools/gpu_step_events.py 16 ref > $OUT/events_shipped.txt 2>&1
