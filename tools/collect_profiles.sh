#!/bin/bash
# run ON THE GPU BOX (gpurun): rocprofv3 kernel statistics + HBM-side traffic counters + SQ counters of `bench.py` at 256^3.
# Counters go in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with other trace domains).
# bash tools/collect_profiles.sh <tag> [bench arguments, e.g. --workload ppb_stretched]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprofv3 options ..., then `--`, then bench options
    local name=$1; shift
    timeout -k 10 240 rocprofv3 "$@" > $R/gpurun_out/${TAG}_$name.log 2>&1 || echo "pass $name failed"
    echo "pass $name done"
}
run stats --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@"
run fetch --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@"
run write --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@"
run valu --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_valu -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@"
run mix --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 --output-format csv -d $R/gpurun_out/${TAG}_mix -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@"
grep -h '"metric"' $R/gpurun_out/${TAG}_stats.log | cut -c1-200
