#!/bin/bash
# Copies the summaries of a tools/profile_round.sh run (gpurun_out/<tag>/, scratch) into profiles/ (tracked) under the round's
# names.  usage (this container, repo root): bash tools/collect_profiles.sh r04
TAG=${1:-r05}
S=gpurun_out/$TAG; D=profiles
cp $S/stats/bench_kernel_stats.csv $D/${TAG}_bench_kernel_stats.csv
cp $S/stats_msm/bench_kernel_stats.csv $D/${TAG}_bench_msm_only_kernel_stats.csv
cp $S/stats_msm_depth1/bench_kernel_stats.csv $D/${TAG}_bench_msm_only_depth1_kernel_stats.csv
cp $S/op_rates.txt $D/${TAG}_op_rates.txt
cp $S/clock_probe.txt $D/${TAG}_clock_probe.txt
cp $S/traffic_$TAG.json $D/traffic_$TAG.json
cp $S/valu_model_$TAG.json $D/valu_model_$TAG.json
cp $S/prove_step_timeline.txt $D/${TAG}_prove_step_timeline.txt
cp $S/prove_step_events.txt $D/${TAG}_prove_step_events.txt
cp $S/prove_step_valu_per_kernel.txt $D/${TAG}_prove_step_valu_per_kernel.txt
python3 tools/pmc_sum.py $S/valu > $D/${TAG}_msm_valu_per_kernel.txt 2>/dev/null
cp $S/bench_line.json $D/${TAG}_bench_line.json
[ -f $S/bench_detail.json ] && cp $S/bench_detail.json $D/${TAG}_bench_detail.json
[ -f $S/bench_rehearse.json ] && cp $S/bench_rehearse.json $D/${TAG}_bench_line_collective_rehearsal.json
ls -la $D | grep ${TAG}_ | wc -l
