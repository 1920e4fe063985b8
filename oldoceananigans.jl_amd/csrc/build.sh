#!/bin/bash
# Builds libocn_mi355x.so for gfx950 in-tree (next to this script). hipcc cross-compiles without a GPU.
#   -ffp-contract=off : FMAs only where the reference writes @muladd / fma (see ocn_device.h)
#   -fhip-fp32-correctly-rounded-divide-sqrt : exact Float32 reciprocal in newton_div
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${OCN_OUT:-$here/libocn_mi355x.so}"   # OCN_OUT / OCN_EXTRA_FLAGS: alternative builds for A/B timing (tools/)
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
    "$here/ocn_api.hip" -o "$out" \
    -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib ${OCN_EXTRA_FLAGS:-}
echo "built $out"
