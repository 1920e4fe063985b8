#!/bin/bash
# ON THE GPU BOX: one rocprofv3 --pmc pass over bench.py, per-kernel averages of the counters for kernels whose name contains <pattern>.
# bash tools/pmc_kernel.sh <tag> <pattern> "<counters>" [bench args]     (counters in their own pass: no trace domains, see MI355X_MICROARCH.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; PAT=$2; CNT=$3; shift 3
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/${TAG}_pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_pmc.log 2>&1 || echo "rocprofv3 failed"
python3 - "$R/gpurun_out/${TAG}_pmc" "$PAT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k[:70], {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
