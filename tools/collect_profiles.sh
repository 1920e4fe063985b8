#!/bin/bash
# run ON THE GPU BOX (gpurun): rocprofv3 kernel statistics + HBM-side traffic counters of `bench.py` at 256^3.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with other trace domains).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_valu -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_valu.log 2>&1
grep -h '"metric"' $R/gpurun_out/${TAG}_stats.log | cut -c1-160
