R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r4
cp $R/vdf_amd/libvdf_hip.so /tmp/new.so
cp $R/ab/libvdf_hip_v1.so $R/vdf_amd/libvdf_hip.so
timeout -k 10 200 python3 $R/tools/gpu_step_events.py 16 ref > $R/gpurun_out/r4/events_v1.txt 2>&1
cp /tmp/new.so $R/vdf_amd/libvdf_hip.so
timeout -k 10 200 python3 $R/tools/gpu_step_events.py 16 ref > $R/gpurun_out/r4/events_new.txt 2>&1
