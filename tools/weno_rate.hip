// Pure-VALU rate of the WENO-5 face flux (ocn_device.h, same flags as the library) on register data: ns per flux per SIMD and the
// in-kernel clock (s_memtime / s_memrealtime). hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//   tools/weno_rate.hip -o tools/_bin/weno_rate
#include "../oldoceananigans.jl_amd/csrc/ocn_device.h"
#include <cstdio>
template <int MODE, int UNROLL> __global__ void __launch_bounds__(1024) k(double *out, unsigned long long *stamps, int iters, double seed) {
    double s0 = seed + 0.001 * threadIdx.x, s1 = s0 * 1.1, s2 = s0 * 0.9, s3 = s0 * 1.3, s4 = s0 * 0.7, s5 = s0 * 1.05;
    double q0 = 0.3 + s0, q1 = 0.2 - s1, q2 = 0.1 + s2, q3 = -0.2 + s3;
    double acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll UNROLL
    for (int it = 0; it < iters; ++it) {
        double ut, F;
        if (MODE == 0) {            // momentum flux: symmetric interpolation + biased WENO
            ut = symmetric_interp(1.5 * q0, 1.5 * q1, 1.5 * q2, 1.5 * q3, false, 5, false, 100);
            F = ut * weno5_biased(s0, s1, s2, s3, s4, s5, ut > 0);
        } else if (MODE == 1) {     // left-biased only (no operand selection)
            ut = symmetric_interp(1.5 * q0, 1.5 * q1, 1.5 * q2, 1.5 * q3, false, 5, false, 100);
            F = ut * weno5_biased(s0, s1, s2, s3, s4, s5, true);
        } else {                    // tracer flux
            ut = q2;
            F = 1.5 * ut * weno5_biased(s0, s1, s2, s3, s4, s5, ut > 0);
        }
        acc += F;
        s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s5 * 0.999 + 1e-3 * F;
        q0 = q1; q1 = q2; q2 = q3; q3 = -q3 * 0.99 + 1e-3 * F;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 12345.678) out[0] = acc;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
}
template <int MODE, int UNROLL> void run(const char *name, double *d, unsigned long long *st) {
    for (int wps : {1, 2, 4}) {
        const int iters = 4000, nw = 256 * 4 * wps;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(256), dim3(256 * wps), 0, 0, d, st, 100, 1.0);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(256), dim3(256 * wps), 0, 0, d, st, iters, 1.0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long h[2 * 4096];
        (void)hipMemcpy(h, st, 16 * nw, hipMemcpyDeviceToHost);
        double ghz = 0; for (int w = 0; w < nw; ++w) ghz += (double)h[2 * w] / (double)h[2 * w + 1] * 0.1;
        ghz /= nw;
        printf("%-26s waves/SIMD %d: %.3f ms  %.1f ns per flux per SIMD = %.1f cycles at the in-kernel clock %.2f GHz  (chip: %.1f Gflux/s)\n", name, wps, ms,
               ms * 1e6 / ((double)iters * wps), ms * 1e6 / ((double)iters * wps) * ghz, ghz, 1024.0 * 64 * iters * wps / (ms * 1e6));
    }
}
int main() {
    double *d; unsigned long long *st; (void)hipMalloc(&d, 64); (void)hipMalloc(&st, 16 * 4096);
    run<0, 1>("momentum flux", d, st); run<1, 1>("momentum flux, left only", d, st); run<2, 1>("tracer flux", d, st);
    run<0, 8>("momentum flux, 8 x unrolled", d, st); run<0, 32>("momentum flux, 32 x unrolled", d, st);
    return 0;
}
