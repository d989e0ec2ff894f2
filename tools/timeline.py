"""Print the kernel timeline of the last complete prove_step from a rocprofv3 rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_step_z"
rows = list(db.execute("select name,start,end,stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][1]
for r in rows[i0:i1 + 1]:
    nm = r[0].split('(')[0].replace('void vdf::', '').replace('vdf::', '')
    print(f"{(r[1]-t0)/1000:8.1f} {(r[2]-t0)/1000:8.1f} {(r[2]-r[1])/1000:7.1f} s{r[3]} {nm[:44]}")
