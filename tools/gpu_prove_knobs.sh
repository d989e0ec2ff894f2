# steady-state prove_step rate under each scheduling knob, same box, same run: bash tools/gpu_prove_knobs.sh
cd $GRAFT_REPO_ROOT
for e in "X=0" "VDF_NOVA_LOOKAHEAD_EARLY=0" "VDF_NOVA_T_AHEAD=1" "VDF_NOVA_T_AHEAD=0" "VDF_NOVA_NIFS_AHEAD=0" "VDF_MSM_DIRECT=0" "VDF_NOVA_SEQ_SYNTH=1" "X=1"; do
  echo "$e: $(env $e python tools/gpu_prove_time.py 16 40 | tail -n 2 | head -n 1)"
done
