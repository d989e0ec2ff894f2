"""Per-kernel average duration from a rocprofv3 rocpd database (kernel-trace)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 from kernels group by name order by 4 desc"))
for r in rows:
    nm = r[0].split('(')[0].replace('void vdf::', '').replace('vdf::', '')
    if 'at::' in nm or 'rocclr' in nm: continue
    print(f"{nm[:44]:44s} calls {r[1]:4d}  avg {r[2]:9.1f} us")
