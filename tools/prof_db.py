"""Per-kernel summary of a rocprofv3 rocpd database: python tools/prof_db.py <results.db> [rows]  (count, average, minimum, total per step)"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s "
                 "on d.kernel_id=s.id group by 1").fetchall()
rows.sort(key=lambda r: -r[4])
tot = sum(r[4] for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{r[0][:84]:84s} n={r[1]:5d} avg={r[2]/1e3:8.1f} min={r[3]/1e3:8.1f} us  {100*r[4]/tot:5.1f} %")
print(f"total kernel time {tot/1e6:.2f} ms")
