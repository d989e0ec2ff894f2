# The proof tests under every tuning switch (results must not depend on scheduling): bash tools/gpu_env_matrix.sh ["VAR=val" ...]
# (no arguments: the whole list, ~90 s per switch -- more than one gpurun call allows: pass a part of the list per call)
# Each run is a process of its own: the environment is read ONCE per process into the tuning structs (vdf_hip_tuning, vdf_nova_tuning);
# tests/test_gpu_tuning.py drives the same switches through the structs inside one process.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LIST=("$@")
[ ${#LIST[@]} -eq 0 ] && LIST=("VDF_MSM_DIRECT=0" "VDF_NOVA_DIGIT_WINDOW=0" "VDF_NOVA_T_AHEAD=0" "VDF_NOVA_T_AHEAD=1" "VDF_NOVA_NIFS_AHEAD=0" "VDF_NOVA_LOOKAHEAD_EARLY=0" \
         "VDF_NOVA_SEQ_SYNTH=1" "VDF_NOVA_DIGIT_WINDOW=8" \
         "VDF_NOVA_PACKED_COMMIT=0" "VDF_MSM_DIRECT_FUSED=0" "VDF_NOVA_GATE=0" "VDF_NOVA_FOLD_ON_ROWS=0" "VDF_NOVA_T_PARTS=2" "VDF_NIFS_LANES=1" \
         "VDF_NIFS_LANES=4" "VDF_MSM_LIGHT_PRIO=1" "VDF_MSM_DIRECT_PRIO=0" "VDF_NOVA_LOOKAHEAD_PRIO=3" "VDF_HOST_NO_ADX=1" "VDF_NOVA_STENCIL=0" "VDF_NOVA_SIDE_ACC_WG=2" "VDF_MSM_ACC_WG=3" "VDF_NOVA_DIGIT_BUDGET_GIB=72" "VDF_NIFS_FUSED=0" "VDF_NOVA_COMPRESS_QUEUES=0" \
         "VDF_MSM_FIXUP_SERIAL=0" "VDF_MSM_SORT_STAGED=0" "VDF_MSM_GLV=0" "VDF_NOVA_FOLD_FUSED=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=4")
for e in "${LIST[@]}"; do
  env $e timeout -k 10 400 python -m pytest tests/test_gpu_nova.py tests/test_gpu_compress.py -x -q -k "not baseline_sizes and not config_5 and not full_size" > gpurun_out/env_test.log 2>&1; echo "$e: $(tail -n 1 gpurun_out/env_test.log)"
done
