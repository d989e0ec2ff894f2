cd $GRAFT_REPO_ROOT
for e in "VDF_MSM_DIRECT=0" "VDF_NOVA_DIGIT_WINDOW=0" "VDF_NOVA_T_AHEAD=0" "VDF_NOVA_T_AHEAD=1" "VDF_NOVA_NIFS_AHEAD=0" "VDF_NOVA_LOOKAHEAD_EARLY=0" "VDF_NOVA_SEQ_SYNTH=1" "VDF_NOVA_DIGIT_WINDOW=8"; do
  env $e timeout -k 10 300 python -m pytest tests/test_gpu_nova.py tests/test_gpu_compress.py -x -q -k "not baseline_sizes" > gpurun_out/env_test.log 2>&1; echo "$e: $(tail -n 1 gpurun_out/env_test.log)"
done
