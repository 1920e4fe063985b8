// v_cndmask_b32 cost on gfx950 in the patterns the WENO flux uses. hipcc --offload-arch=gfx950 -O2 tools/valu_rates2.hip -o tools/_bin/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int OP> __global__ void __launch_bounds__(1024) k(double *out, int iters, double seed) {
    float f0 = (float)seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    float g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0, g6 = 0, g7 = 0;
    double a0 = seed + threadIdx.x, a1 = 0.5 - a0;
    unsigned long long m = 0x5555555555555555ull;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            // e32, vcc constant, dst == src0
            if (OP == 0) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.5f) : "vcc");
            // e32, distinct destination registers
            if (OP == 1) asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %9, %10, vcc\n v_cndmask_b32 %2, %10, %11, vcc\n v_cndmask_b32 %3, %11, %12, vcc\n v_cndmask_b32 %4, %12, %13, vcc\n v_cndmask_b32 %5, %13, %14, vcc\n v_cndmask_b32 %6, %14, %15, vcc\n v_cndmask_b32 %7, %15, %8, vcc" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7) : "vcc");
            // e64 with an SGPR-pair mask
            if (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %8, %9, %16\n v_cndmask_b32_e64 %1, %9, %10, %16\n v_cndmask_b32_e64 %2, %10, %11, %16\n v_cndmask_b32_e64 %3, %11, %12, %16\n v_cndmask_b32_e64 %4, %12, %13, %16\n v_cndmask_b32_e64 %5, %13, %14, %16\n v_cndmask_b32_e64 %6, %14, %15, %16\n v_cndmask_b32_e64 %7, %15, %8, %16" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7), "s"(m));
            // the flux pattern: one FP64 compare into vcc, then 8 selects (counts as 9 instructions, reported per 8)
            if (OP == 3) asm volatile("v_cmp_lt_f64 vcc, %16, %17\n v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %9, %10, vcc\n v_cndmask_b32 %2, %10, %11, vcc\n v_cndmask_b32 %3, %11, %12, vcc\n v_cndmask_b32 %4, %12, %13, vcc\n v_cndmask_b32 %5, %13, %14, vcc\n v_cndmask_b32 %6, %14, %15, vcc\n v_cndmask_b32 %7, %15, %8, vcc" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7), "v"(a0), "v"(a1) : "vcc");
            // v_mov_b32 for comparison
            if (OP == 4) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %10\n v_mov_b32 %3, %11\n v_mov_b32 %4, %12\n v_mov_b32 %5, %13\n v_mov_b32 %6, %14\n v_mov_b32 %7, %15" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
            // selects interleaved with FP64 multiplies (4 + 4)
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_mul_f64 %16, %16, %17\n v_cndmask_b32 %1, %9, %10, vcc\n v_mul_f64 %16, %16, %17\n v_cndmask_b32 %2, %10, %11, vcc\n v_mul_f64 %16, %16, %17\n v_cndmask_b32 %3, %11, %12, vcc\n v_mul_f64 %16, %16, %17" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7), "v"(a0), "v"(a1) : "vcc");
        }
    }
    long long t1 = clock64();
    double s = a0 + a1 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7;
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (double)(t1 - t0);
}
template <int OP> void run(const char *name, double *d) {
    for (int wps : {1, 2, 4}) {
        const int iters = 2000;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * 4 * wps), 0, 0, d, 10, 1.0);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * 4 * wps), 0, 0, d, iters, 1.0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double ninstr = (double)iters * REP;
        printf("%-34s waves/SIMD %d: %.3f ms  %.2f ns per instr per SIMD; wave 0 alone-equivalent %.2f ticks per instr\n", name, wps, ms, ms * 1e6 / (ninstr * wps), h[1] / ninstr);
    }
}
int main() {
    double *d; (void)hipMalloc(&d, 64);
    run<4>("v_mov_b32", d); run<0>("cndmask e32 vcc dst=src0", d); run<1>("cndmask e32 vcc distinct dst", d); run<2>("cndmask e64 sgpr mask", d);
    run<3>("cmp_lt_f64 + 8 cndmask", d); run<5>("4 cndmask + 4 mul_f64", d);
    return 0;
}
