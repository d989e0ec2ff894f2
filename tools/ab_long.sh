# interleaved A/B, 158 steady-state steps per run: bash tools/ab_long.sh "VAR=val"
for rep in 1 2 3; do for cfg in "" "$1"; do
  echo "[$cfg] $(env $cfg python tools/gpu_prove_time.py 16 160 ref 2>&1 | grep 'steady state')"
done; done
