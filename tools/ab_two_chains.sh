R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r4
cp $R/vdf_amd/libvdf_hip.so /tmp/new.so
run() { echo "== $1"; env $2 timeout -k 10 280 python3 $R/tools/gpu_prove_two_chains.py 16 60 2 2>&1 | grep "chain"; }
run "new, defaults (20 GiB digit budget)" "X=1"
run "new, 72 GiB" "VDF_NOVA_DIGIT_BUDGET_GIB=72"
run "new, 72 GiB, stencil off" "VDF_NOVA_DIGIT_BUDGET_GIB=72 VDF_NOVA_STENCIL=0"
cp $R/ab/libvdf_hip_v1.so $R/vdf_amd/libvdf_hip.so
run "v1 madd, 72 GiB, side fill 2 (round 3's kernel and fill)" "VDF_NOVA_DIGIT_BUDGET_GIB=72 VDF_NOVA_SIDE_ACC_WG=2"
run "v1 madd, 72 GiB, side fill 2, stencil off" "VDF_NOVA_DIGIT_BUDGET_GIB=72 VDF_NOVA_SIDE_ACC_WG=2 VDF_NOVA_STENCIL=0"
cp /tmp/new.so $R/vdf_amd/libvdf_hip.so
