R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/synth; mkdir -p $OUT
cd $R && g++ -O2 -std=c++17 -I include tools/host_synth_bench.cpp -o /tmp/bench_synth -L vdf_amd -lvdf_nova -lvdf_hip -Wl,-rpath,$PWD/vdf_amd && /tmp/bench_synth && (VDF_NOVA_SYNTH_TRACE=1 /tmp/bench_synth 2>&1 | grep "synth side" | sed -n '100,102p;600,602p')
python tools/gpu_prove_time.py 16 12 > $OUT/prove.log 2>&1 || { tail -n 20 $OUT/prove.log; exit 1; }
tail -n 3 $OUT/prove.log
