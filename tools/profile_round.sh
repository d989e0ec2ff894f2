# Round profile: kernel stats of the default bench run + the two HBM traffic passes (separate --pmc runs, as
# MI355X_MICROARCH.md prescribes).  Run on the GPU box from the repo root: bash tools/profile_round.sh r01
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 > $OUT/bench_under_rocprof.log 2>&1 || exit 1
# the MSM leg alone, as timed (two steps in flight): k_accumulate's average here pairs with roofline.avg_launch_ms
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_msm -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-prove --no-cpu > $OUT/bench_msm_under_rocprof.log 2>&1 || exit 1
# one step at a time: pairs with roofline.isolated_launch_ms
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_msm_depth1 -o bench -- python $R/bench.py --depth 1 --steps 20 --warmup 3 --no-prove --no-cpu > $OUT/bench_msm_depth1_under_rocprof.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python $R/bench.py --depth 1 --steps 3 --warmup 1 --no-prove --no-cpu > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python $R/bench.py --depth 1 --steps 3 --warmup 1 --no-prove --no-cpu > $OUT/write.log 2>&1 || exit 1
cd $R && timeout -k 10 500 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err || exit 1
tail -c 600 $OUT/bench_line.json
