"""copy a rocprofv3 kernel-stats summary that tools/prof_stats.sh left under gpurun_out/ into profiles/ with the header lines the committed
profiles carry (command, bench line of the same run):  python tools/commit_profile.py <tag> <profiles/name.csv> "<command as run>" """
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
f = max(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # the newest collection
line = None
for ln in open(os.path.join(ROOT, "gpurun_out", tag + "_stats.log")):
    if ln.startswith('{"metric"'):
        line = json.loads(ln)
rows = list(csv.DictReader(open(f)))
with open(os.path.join(ROOT, out), "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}  (MI355X; 10 timed + 10 individually synchronised steps after 2 + 2 warm-up steps)\n")
    if line:
        o.write(f"# bench line of the same run: ms_per_step {line['ms_per_step']:.3f} (median {line.get('ms_per_step_median', float('nan')):.3f}), "
                f"roofline.avg_launch_ms {line['roofline']['avg_launch_ms']:.4f}\n")
    o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows:
        o.write('"%s",%s,%s,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
print("wrote", out, "ms_per_step", line and line["ms_per_step"])
