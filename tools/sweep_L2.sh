run() { echo "== n=$1 L=$2"; VDF_MSM_L=$2 timeout -k 10 120 python tools/gpu_msm_time.py n$1 tbl16x1 2>&1 | grep -E "stages" || exit 1; }
run 98304 8
run 196608 16
run 393216 32
run 786432 64
run 1572864 128
run 131072 16
run 65536 16
run 262144 32
run 131072 32
