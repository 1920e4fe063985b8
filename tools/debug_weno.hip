// debug: compare device WENO stages with host (gcc, -ffp-contract=off) evaluation of the same expressions
#include "../oldoceananigans.jl_amd/csrc/ocn_device.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
extern "C" double oro_weno5_biased(const double S[6], int left);
extern "C" double oro_newton_div_f32(double a, double b);
extern "C" void host_stages(const double *S, int left, double *out);

__global__ void k(const double *S, int n, double *out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double *s = S + 6 * t;
    bool left = t & 1;
    double a0 = left ? s[2] : s[3], a1 = left ? s[3] : s[2], a2 = left ? s[4] : s[1];
    double b0 = left ? s[1] : s[4];
    double c0 = left ? s[0] : s[5];
    double be0 = beta3(a0, a1, a2, 10, -31, 11, 25, -19, 4);
    double be1 = beta3(b0, a0, a1, 4, -13, 5, 13, -13, 4);
    double be2 = beta3(c0, b0, a0, 4, -19, 11, 25, -31, 10);
    double tau = fabs(be0 - be2);
    double r0 = newton_div_f32(tau, be0 + OCN_WENO_EPS);
    float bl = (float)(be0 + OCN_WENO_EPS);
    float inv = 1.0f / bl;
    double al0 = OCN_W3C0 * (1.0 + r0 * r0);
    double sinv = 1.0 / (al0 + 0.37);
    double *o = out + 8 * t;
    o[0] = be0; o[1] = be1; o[2] = be2; o[3] = r0; o[4] = (double)inv; o[5] = al0; o[6] = sinv;
    o[7] = weno5_biased(s[0], s[1], s[2], s[3], s[4], s[5], left);
}
int main() {
    int n = 1 << 16;
    std::vector<double> S(6 * n), out(8 * n), ref(8 * n);
    srand(1);
    for (auto &x : S) x = (rand() / (double)RAND_MAX - 0.5) * 3;
    double *dS, *dO;
    hipMalloc(&dS, S.size() * 8); hipMalloc(&dO, out.size() * 8);
    hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dS, n, dO);
    hipMemcpy(out.data(), dO, out.size() * 8, hipMemcpyDeviceToHost);
    int bad[8] = {0};
    for (int t = 0; t < n; ++t) {
        host_stages(&S[6 * t], t & 1, &ref[8 * t]);
        for (int q = 0; q < 8; ++q) if (out[8 * t + q] != ref[8 * t + q]) { if (bad[q]++ < 2) printf("stage %d t %d gpu %.17g host %.17g\n", q, t, out[8*t+q], ref[8*t+q]); }
    }
    for (int q = 0; q < 8; ++q) printf("stage %d mismatches %d / %d\n", q, bad[q], n);
    return 0;
}
