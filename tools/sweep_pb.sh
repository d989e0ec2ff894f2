for pb in 7 8 9 10; do
  echo "== 2^20 pb=$pb"; VDF_MSM_PB=$pb timeout -k 10 120 python tools/gpu_msm_time.py 20 tbl16x1 2>&1 | grep -E "stages|parity" || exit 1
done
for pb in 6 7 8 9; do
  echo "== 2^18 pb=$pb"; VDF_MSM_PB=$pb timeout -k 10 120 python tools/gpu_msm_time.py 18 tbl16x1 2>&1 | grep -E "stages|parity" || exit 1
done
