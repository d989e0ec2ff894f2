R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/early; mkdir -p $OUT
cd $R && g++ -O2 -std=c++17 -I include tools/scratch/robench.cpp -o /tmp/robench -L vdf_amd -lvdf_nova -lvdf_hip -Wl,-rpath,$PWD/vdf_amd && /tmp/robench
g++ -O3 -std=c++17 -I vdf_amd/csrc/host -I include -I vdf_amd/csrc tools/scratch/mulbench.cpp vdf_amd/csrc/host/host_math.cpp -o /tmp/mulbench && /tmp/mulbench
VDF_NOVA_SYNTH_TRACE=1 python tools/gpu_prove_time.py 16 12 > $OUT/prove_trace.log 2>&1 || { tail -n 20 $OUT/prove_trace.log; exit 1; }
grep "synth side" $OUT/prove_trace.log | tail -n 4; tail -n 2 $OUT/prove_trace.log
