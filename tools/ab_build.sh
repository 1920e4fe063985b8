#!/bin/bash
# Build alternative libraries for A/B timing on the GPU box: bash tools/ab_build.sh tagA "-DFOO=1" tagB "-DFOO=0" ...
# -> tools/_bin/lib_<tag>.so (travels with gpurun); run with OCN_LIB=tools/_bin/lib_<tag>.so python tools/tune_roles.py
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$here/_bin"
while [ $# -ge 2 ]; do
    ( OCN_OUT="$here/_bin/lib_$1.so" OCN_EXTRA_FLAGS="$2" bash "$here/../oldoceananigans.jl_amd/csrc/build.sh" 2>&1 | grep -E "error|built" ) &
    shift 2
done
wait
