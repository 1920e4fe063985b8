// ocn_tendency_fused.h -- fused, flux-sharing WENO-5 tendency kernel (all of Gu, Gv, Gw, Gc in one pass).
#pragma once
#include "ocn_device.h"

static inline bool fused_tendency_supported(const DGrid &, const int *) { return false; }
static inline int launch_fused_tendency(const DGrid &, hipStream_t, const double *, const double *, const double *,
                                        const double *const *, int, double *, double *, double *, double *const *) {
    return -2;
}
