#!/bin/bash
# ON THE GPU BOX: one rocprofv3 --kernel-trace --stats pass of bench.py; prints the top kernels. bash tools/prof_stats.sh <tag> [ENV=VAL ...] -- [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
while [ "$1" != "--" ] && [ -n "$1" ]; do export "$1"; shift; done
[ "$1" == "--" ] && shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_stats.log 2>&1 || echo "rocprofv3 failed"
f=$(find $R/gpurun_out/${TAG}_stats -name "*kernel_stats.csv" | head -1)
grep -h '"metric"' $R/gpurun_out/${TAG}_stats.log | cut -c1-160
head -32 "$f" | cut -c1-200
