#!/bin/bash
# ON THE GPU BOX: rocprofv3 --kernel-trace --stats over a python script (not bench.py); prints per-kernel averages.  bash tools/prof_script.sh <tag> <script> [args ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
mkdir -p $R/gpurun_out
S=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $S "$@" > $R/gpurun_out/${TAG}_stats.log 2>&1 || echo "rocprofv3 failed"
tail -3 $R/gpurun_out/${TAG}_stats.log
python3 - "$R/gpurun_out/${TAG}_stats" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print("%-56s %6s %10.1f us  %5.1f %%" % (r["Name"].split("(")[0][:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
