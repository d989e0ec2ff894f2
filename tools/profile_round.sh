# Round profile, run on the GPU box from the repo root:  bash tools/profile_round.sh r02
# Kernel stats of the default bench run, the MSM leg as timed and one step at a time, the two HBM traffic passes and the
# VALU count (separate --pmc passes, kernel trace only, as MI355X_MICROARCH.md prescribes), and the raw micro-benchmark
# logs the issue-rate model rests on.  Every bench.py under rocprofv3 runs with --no-cpu: nothing is spawned under the
# profiler (the cpu_baseline leg may rebuild the C restatement) and no CPU MSM sits inside a profiled run.
TAG=${1:-r05}
COMMIT=${2:-unknown}      # the commit of the code being profiled (the GPU box has no .git): stamped into the JSON files
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
make -s -C $R/oracle all || exit 1
cd /tmp && export TMPDIR=/tmp
for t in op_rates clock_probe; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I $R/vdf_amd/csrc $R/tools/ubench/$t.hip -o $OUT/$t || exit 1
  timeout -k 10 120 $OUT/$t > $OUT/$t.txt 2>&1 || exit 1
done
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-sizes > $OUT/bench_under_rocprof.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_msm -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-prove --no-cpu --no-sizes > $OUT/bench_msm_under_rocprof.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_msm_depth1 -o bench -- python $R/bench.py --depth 1 --steps 20 --warmup 3 --no-prove --no-cpu --no-sizes > $OUT/bench_msm_depth1_under_rocprof.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $R/bench.py --depth 1 --steps 3 --warmup 1 --no-prove --no-cpu --no-sizes > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $R/bench.py --depth 1 --steps 3 --warmup 1 --no-prove --no-cpu --no-sizes > $OUT/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $OUT/valu -o p -- python $R/bench.py --depth 1 --steps 3 --warmup 1 --no-prove --no-cpu --no-sizes > $OUT/valu.log 2>&1 || exit 1
# one steady-state prove_step: timeline and VALU per kernel
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/prove -o p -- python $R/tools/gpu_prove_time.py 16 10 ref > $OUT/prove.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $OUT/prove_valu -o p -- python $R/tools/gpu_prove_time.py 16 8 ref > $OUT/prove_valu.log 2>&1 || exit 1
cd $R
python tools/parse_pmc.py $OUT/fetch $OUT/write $OUT/traffic_$TAG.json k_accumulate 15728640 $COMMIT > $OUT/traffic.log 2>&1
python tools/make_valu_model.py $OUT/op_rates.txt $OUT/clock_probe.txt $OUT/valu 15728640 $OUT/valu_model_$TAG.json $TAG $COMMIT > $OUT/valu_model.log 2>&1
python tools/timeline.py $(ls $OUT/prove/*.db | head -1) k_nifs_cross 7 3 > $OUT/prove_step_timeline.txt 2>&1
python tools/gpu_step_events.py 16 ref > $OUT/prove_step_events.txt 2>&1
python tools/pmc_sum.py $OUT/prove_valu > $OUT/prove_step_valu_per_kernel.txt 2>&1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err || exit 1
cp $R/bench_detail.json $OUT/bench_detail.json
tail -c 800 $OUT/bench_line.json
