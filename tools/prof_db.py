"""summarise a rocprofv3 results .db (kernel-trace): per-kernel count / total / average, as CSV on stdout.
usage: python tools/prof_db.py gpurun_out/prof/x_results.db [steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"')
for r in rows:
    print(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100 * r[2] / tot:.2f}')
if steps:
    print(f'# sum of kernel time per step: {tot / steps / 1e6:.3f} ms', file=sys.stderr)
