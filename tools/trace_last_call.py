"""Timeline of the kernels of the LAST MSM call in a rocprofv3 kernel trace (csv): start (ms from the call's first kernel), duration, queue.
   python3 tools/trace_last_call.py <kernel_trace.csv> [first-kernel substring: k_part]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "k_part<"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call starts at the last k_part<..., false> that follows a k_point_sum / k_final (or the start)
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"] and "false" in r["Kernel_Name"]]
# two halves per split call: take the second-to-last 'false' launch whose predecessor set contains the sum
first = starts[-2] if len(starts) >= 2 and "--split" in sys.argv else starts[-1]
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    print("%9.3f %9.3f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
