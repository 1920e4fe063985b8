#!/bin/bash
# one rocprofv3 --pmc pass over tools/run_tendency.py (GPU box): bash tools/pmc_pass.sh <tag> <impl> <counters...>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
IMPL=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/tools/run_tendency.py 256 $IMPL 7 0 2 3 > $R/gpurun_out/pmc_$TAG.log 2>&1
python3 - << PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_$TAG/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tendency_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
