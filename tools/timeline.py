"""Print a kernel timeline from a rocprofv3 rocpd database: the span between two occurrences of an anchor kernel
(default: k_nifs_cross, which runs twice per prove_step -- secondary side, then primary side -- so the span is one
steady-state step with its overlaps)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_nifs_cross"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = list(db.execute("select name,start,end,stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
span = int(sys.argv[4]) if len(sys.argv) > 4 else 2
i0, i1 = idx[-back], idx[-back + span]    # cross terms per step: secondary side, primary side (+ the early rows of T when they run apart)
t0 = rows[i0][1]
for r in rows[i0:i1 + 1]:
    nm = r[0].split('(')[0].replace('void vdf::', '').replace('vdf::', '')
    print(f"{(r[1]-t0)/1000:8.1f} {(r[2]-t0)/1000:8.1f} {(r[2]-r[1])/1000:7.1f} s{r[3]} {nm[:44]}")
