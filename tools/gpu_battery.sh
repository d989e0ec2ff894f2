# one-off robustness battery on the GPU box (about ten minutes): bash tools/gpu_battery.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/battery; mkdir -p $O
timeout -k 10 900 python tools/gpu_msm_fuzz.py 2500 23 > $O/msm_fuzz.log 2>&1; echo "msm fuzz: $(tail -n 1 $O/msm_fuzz.log)"
timeout -k 10 600 python tools/gpu_nova_fuzz.py 60 31 > $O/nova_fuzz.log 2>&1; echo "nova fuzz: $(tail -n 1 $O/nova_fuzz.log)"
timeout -k 10 600 python tools/gpu_prove_two_chains.py 16 300 2 > $O/two_chains.log 2>&1; echo "two chains x 300 steps: $(tail -n 3 $O/two_chains.log | tr '\n' ' ')"
timeout -k 10 600 python tools/gpu_prove_soak.py > $O/soak.log 2>&1; echo "soak: $(tail -n 1 $O/soak.log)"
timeout -k 10 600 python tools/gpu_wire_fuzz.py 600 5 > $O/wire_fuzz.log 2>&1; echo "wire fuzz: $(tail -n 1 $O/wire_fuzz.log)"
