cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_compress
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_compress -o p -- python $GRAFT_REPO_ROOT/tools/gpu_compress_time.py 16 > $GRAFT_REPO_ROOT/gpurun_out/prof_compress.log 2>&1
tail -n 3 $GRAFT_REPO_ROOT/gpurun_out/prof_compress.log
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_compress/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), ("%.2f ms" % (float(r["TotalDurationNs"]) / 1e6)).rjust(10), r["Percentage"])
PY
