import sqlite3, sys
for d in sys.argv[2:]:
    db = sqlite3.connect(d)
    q = "select counter_name, count(*), avg(value) from counters_collection where kernel_name like ? group by counter_name"
    for r in db.execute(q, ("%" + sys.argv[1] + "%",)): print(f"{r[0]:28s} launches {r[1]:3d} avg {r[2]:.0f}")
