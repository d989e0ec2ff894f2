# HBM traffic of the early rows' cross term, stencil kernel against the generic sparse kernel (tools/gpu_stencil_time.py), from
# the FETCH_SIZE / WRITE_SIZE counters (two separate --pmc passes, kernel trace only, as MI355X_MICROARCH.md prescribes; KiB units).
# usage (GPU box, repo root): bash tools/pmc_stencil.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_stencil
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python3 $R/tools/gpu_stencil_time.py 16 > $OUT/$c.log 2>&1 || exit 1
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, cnt = collections.Counter(), collections.Counter()
    for f in glob.glob(sys.argv[1] + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            nm = r["Kernel_Name"].split("(")[0].replace("void vdf::", "")
            if "nifs_cross" not in nm: continue
            tot[nm] += float(r["Counter_Value"]); cnt[nm] += 1
    for nm in tot: res[nm][c] = tot[nm] / cnt[nm] * 1024.0     # KiB -> bytes per launch
for nm, d in sorted(res.items()):
    print("%-44s FETCH_SIZE %7.1f MB  WRITE_SIZE %7.1f MB per launch (as counted; streaming 128-B requests may count double: MI355X_MICROARCH.md)" %
          (nm, d.get("FETCH_SIZE", 0) / 1e6, d.get("WRITE_SIZE", 0) / 1e6))
PY
