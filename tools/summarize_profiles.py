"""turn the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/<tag>_{stats,fetch,write}) into the committed
summaries under profiles/:  <tag>_kernel_stats_256cubed.csv, <tag>_hbm_traffic_256cubed.csv, <tag>_tendency_traffic.json.

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters). On gfx950 FETCH_SIZE under-reports wide coalesced streaming
reads (MI355X_MICROARCH.md, HBM section: half the bytes for 16 B/lane); for this code's 8 B/lane streams the factor is
calibrated on rk3_substep_kernel, whose traffic is known exactly (stage 1: reads 5 fields + 5 tendencies, writes 5 fields
over the 256^3 interior)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
suffix = sys.argv[2] if len(sys.argv) > 2 else "256cubed"      # name of the workload in the output files
N = 256
cells = float(N) ** 3
out = os.path.join(ROOT, "gpurun_out")


def one(pattern):
    files = glob.glob(os.path.join(out, pattern), recursive=True)
    assert files, pattern
    return max(files, key=os.path.getmtime)        # several collections may sit side by side: take the newest


def short(name):
    return name.split("(")[0].strip()


# ---- kernel statistics ------------------------------------------------------------------------------------------------
bench_line = [l for l in open(os.path.join(out, f"{tag}_stats.log")) if l.startswith('{"metric"')]
bench = json.loads(bench_line[0]) if bench_line else {}
rows = list(csv.DictReader(open(one(f"{tag}_stats/**/*kernel_stats.csv"))))
dst = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{suffix}.csv")
with open(dst, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline  (MI355X, 256^3)\n")
    if bench:
        f.write(f"# bench line of the same run: ms_per_step {bench['ms_per_step']:.3f}, roofline.avg_launch_ms "
                f"{bench['roofline']['avg_launch_ms']:.4f} (HIP events on the launch stream) -- compare with AverageNs of the fused kernel below\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows:
        f.write(f"\"{r['Name']}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['Percentage']}\n")
print("wrote", dst)

# ---- traffic counters -------------------------------------------------------------------------------------------------
def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return acc


fetch = per_kernel(one(f"{tag}_fetch/**/*counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(one(f"{tag}_write/**/*counter_collection.csv"), "WRITE_SIZE")
sub = "rk3_substep_kernel"
known_read, known_write = 80.0, 40.0          # B/cell of the stage-1 substep (U, Gn read; U written)
cal_f = known_read / (sum(fetch[sub]) / len(fetch[sub]) / cells) if fetch.get(sub) else 1.6
cal_w = known_write / (sum(write[sub]) / len(write[sub]) / cells) if write.get(sub) else 1.0
dst = os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic_{suffix}.csv")
tend = None
with open(dst, "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --steps 3 --warmup 1, MI355X 256^3\n")
    f.write("# units: counter (KiB) x 1024 B, averaged per launch, divided by 256^3 cells. FETCH_SIZE under-reports 8-B/lane streaming reads on gfx950:\n")
    f.write(f"# calibration on rk3_substep_kernel (known {known_read:.0f} B/cell read, {known_write:.0f} B/cell written): FETCH factor {cal_f:.3f}, WRITE factor {cal_w:.3f}\n")
    f.write("kernel,calls,fetch_B_per_cell_raw,write_B_per_cell_raw,fetch_B_per_cell_calibrated\n")
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        fr = sum(fetch[k]) / len(fetch[k]) / cells
        wr = sum(write[k]) / len(write[k]) / cells if write.get(k) else float("nan")
        f.write(f"\"{k}\",{len(fetch[k])},{fr:.1f},{wr:.1f},{fr * cal_f:.1f}\n")
        if "tendency_kernel<" in k:
            tend = (tend or []) + [(k, len(fetch[k]), sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k]))]
print("wrote", dst)
if tend:
    dst = os.path.join(ROOT, "profiles", f"{tag}_tendency_traffic.json")
    nl = sum(t[1] for t in tend)
    fr, wr = sum(t[1] * t[2] for t in tend) / nl, sum(t[1] * t[3] for t in tend) / nl
    json.dump({"kernel": " + ".join(t[0] for t in tend), "launches": {t[0]: t[1] for t in tend},
               "per_instantiation": {t[0]: {"fetch_bytes_per_launch_raw": t[2], "write_bytes_per_launch_raw": t[3],
                                            "hbm_bytes_per_launch": t[2] * cal_f + t[3]} for t in tend},
               "fetch_bytes_per_launch_raw": fr, "write_bytes_per_launch_raw": wr,
               "fetch_calibration": cal_f, "hbm_bytes_per_launch": fr * cal_f + wr,
               "note": "average over all launches of the run (2 of 3 launches per time-step carry the fused RK3 substep); FETCH_SIZE x "
                       "calibration (rk3_substep_kernel, same 8-B/lane streaming pattern) + WRITE_SIZE; counts Infinity-Cache hits "
                       "too (MI355X_MICROARCH.md HBM section)"}, open(dst, "w"), indent=1)
    print("wrote", dst)

# ---- VALU issue counters (the binding roof of the tendency kernel) ---------------------------------------------------------
vfile = sorted(glob.glob(os.path.join(out, f"{tag}_valu/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
if vfile:
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(vfile[0])):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    k = [k for k in acc if "tendency_kernel<" in k]
    if k:
        merged = defaultdict(list)    # all instantiations (with / without the fused substep): launch-weighted average
        for kk in k:
            for n, v in acc[kk].items():
                merged[n] += v
        c = {n: sum(v) / len(v) for n, v in merged.items()}
        mfile = sorted(glob.glob(os.path.join(out, f"{tag}_mix/**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
        if mfile:                     # instruction classes of the same kernels (a second pass: 8 SQ counters per pass)
            macc = defaultdict(list)
            for r in csv.DictReader(open(mfile[0])):
                if short(r["Kernel_Name"]) in k:
                    macc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            c.update({n: sum(v) / len(v) for n, v in macc.items()})
        dst = os.path.join(ROOT, "profiles", f"{tag}_tendency_valu.json")
        json.dump({"kernel": " + ".join(k), "counters_per_launch": c,
                   "valu_wave_instructions_per_cell": c.get("SQ_INSTS_VALU", 0) * 64 / cells,
                   "note": "rocprofv3 --pmc passes (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY "
                           "SQ_WAIT_INST_ANY SQ_WAVES; instruction classes in a second pass), averaged over the launches of bench.py "
                           "--steps 3 (2 of 3 launches carry the fused RK3 substep); SQ cycle counters are in quad-cycles"},
                  open(dst, "w"), indent=1)
        print("wrote", dst)
