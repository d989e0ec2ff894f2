for n in n196615 n262148; do
for L in 8 12 16 24 32; do
  echo "== $n L=$L"
  VDF_MSM_L=$L timeout -k 10 120 python tools/gpu_msm_time.py $n tbl16x1 2>&1 | grep -E "ms/MSM|stages|parity" || exit 1
done; done
