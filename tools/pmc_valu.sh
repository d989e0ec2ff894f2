# SQ_INSTS_VALU per k_accumulate launch (rocprofv3 --pmc, kernel trace only) -> VALU instructions per bucket addition
# usage (GPU box, repo root): bash tools/pmc_valu.sh <out dir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_valu}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $OUT/valu -o p -- python3 $R/bench.py --depth 1 --steps 3 --warmup 1 --no-prove --no-cpu --no-sizes > $OUT/valu.log 2>&1 || exit 1
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys
vals = []
for f in glob.glob(sys.argv[1] + "/valu/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_accumulate" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
            vals.append(float(r["Counter_Value"]))
vals = [v for v in vals if v >= 0.8 * max(vals)]
per = sum(vals) / len(vals)
print("SQ_INSTS_VALU per k_accumulate launch %.0f over %d launches -> %.1f VALU instructions per bucket addition (15,728,640 entries)" % (per, len(vals), per * 64 / 15728640))
PY
