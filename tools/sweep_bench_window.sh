R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/csweep; mkdir -p $OUT
cd $R
for c in 16 17 18 19 20; do for d in 2 3; do
  timeout -k 10 200 python bench.py --window $c --depth $d --no-prove --no-cpu > $OUT/c${c}_d$d.json 2> $OUT/c${c}_d$d.err || { echo "c=$c d=$d failed"; tail -n 5 $OUT/c${c}_d$d.err; continue; }
  python - <<PY
import json
j=json.load(open("$OUT/c${c}_d$d.json"))
print("c=$c depth=$d", j["value"], j["unit"], j["ms_per_step"], "ms/step  isolated", j["roofline"].get("avg_launch_ms"))
PY
done; done
