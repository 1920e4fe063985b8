"""print the kernel timeline of one RK3 time-step from a rocprofv3 kernel trace (start, end, duration, gap to the previous kernel, queue):
python tools/stage_timeline.py gpurun_out/<tag>_stats [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[-1]
nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 26
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rk3_substep_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
pe = t0
for r in rows[a:a + nrows]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f dur %7.1f gap %7.1f q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, (s - pe) / 1e3, r["Queue_Id"], r["Kernel_Name"].split("(")[0][:60]))
    pe = e
print("step total", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
