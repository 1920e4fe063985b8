// mimic the library's plan usage: keep plans alive, non-blocking stream via hipfftSetStream, Z2Z plan created first
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
__global__ void scale_k(double2 *b, long n, double s) { long q = blockIdx.x * (long)blockDim.x + threadIdx.x; if (q < n) { b[q].x *= s; b[q].y *= s; } }
int main(int argc, char **argv) {
    bool destroy = argc > 1 && atoi(argv[1]);
    bool nullstream = argc > 2 && atoi(argv[2]);
    bool with_c2c = !(argc > 3 && atoi(argv[3]));
    int only = argc > 4 ? atoi(argv[4]) : 0;      // 0: strided and contiguous C2R plans, 1: contiguous only, 2: strided only
    hipStream_t st = 0;
    if (!nullstream) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int sizes[][3] = {{32,16,8},{16,16,16},{8,16,32},{32,32,32},{64,16,8},{32,16,16}};
    for (auto &sz : sizes) for (int strided = 1; strided >= 0; --strided) {
        if ((only == 1 && strided) || (only == 2 && !strided)) continue;
        int Nx = sz[0], Ny = sz[1], Nz = sz[2], Nxh = Nx / 2 + 1;
        size_t n = (size_t)Nx * Ny * Nz, nh = (size_t)Nxh * Ny * Nz;
        std::vector<double> h(n);
        srand(1);
        for (auto &x : h) x = rand() / (double)RAND_MAX - 0.5;
        int H = 3, Px = Nx + 2 * H, Py = Ny + 2 * H, Pz = Nz + 2 * H;
        size_t np = (size_t)Px * Py * Pz;
        double *din, *dout; hipfftDoubleComplex *dc, *dstor;
        hipMalloc(&dstor, n * 16); hipMalloc(&din, n * 8); hipMalloc(&dc, nh * 16); hipMalloc(&dout, np * 8);
        hipMemcpyAsync(din, h.data(), n * 8, hipMemcpyHostToDevice, st);
        hipMemsetAsync(dout, 0, np * 8, st);
        hipfftHandle c, f, b;
        int n3[3] = {Nz, Ny, Nx};
        if (with_c2c) { hipfftPlan3d(&c, Nz, Ny, Nx, HIPFFT_Z2Z); hipfftSetStream(c, st); }
        hipfftPlanMany(&f, 3, n3, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, 1);
        int ie[3] = {Nz, Ny, Nxh}, oe[3] = {Pz, Py, Px};
        if (strided) hipfftPlanMany(&b, 3, n3, ie, 1, (int)nh, oe, 1, (int)np, HIPFFT_Z2D, 1);
        else hipfftPlanMany(&b, 3, n3, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, 1);
        hipfftSetStream(f, st); hipfftSetStream(b, st);
        hipfftExecD2Z(f, din, dc);
        scale_k<<<(nh + 255) / 256, 256, 0, st>>>((double2 *)dc, (long)nh, 1.0 / n);
        double *o = strided ? dout + H + (size_t)Px * (H + (size_t)Py * H) : din;
        hipfftExecZ2D(b, dc, o);
        hipStreamSynchronize(st);
        double err = 0;
        if (strided) {
            std::vector<double> p(np); hipMemcpy(p.data(), dout, np * 8, hipMemcpyDeviceToHost);
            for (int k = 0; k < Nz; ++k) for (int j = 0; j < Ny; ++j) for (int i = 0; i < Nx; ++i)
                err = fmax(err, fabs(p[(i + H) + (size_t)Px * ((j + H) + (size_t)Py * (k + H))] - h[i + (size_t)Nx * (j + (size_t)Ny * k)]));
        } else {
            std::vector<double> out(n); hipMemcpy(out.data(), din, n * 8, hipMemcpyDeviceToHost);
            for (size_t q = 0; q < n; ++q) err = fmax(err, fabs(out[q] - h[q]));
        }
        printf("%dx%dx%d strided %d err %.2e\n", Nx, Ny, Nz, strided, err);
        if (!strided || only == 2) {
            // the unit-stride batched 1-D complex plans of the library's per-direction ("general") path, created while everything above
            // is alive: round trip over the whole array for each line length
            int lens[3] = {Nx, Ny, Nz};
            for (int d = 0; d < 3; ++d) {
                hipfftHandle l;
                int nn[1] = {lens[d]};
                hipfftPlanMany(&l, 1, nn, nullptr, 1, lens[d], nullptr, 1, lens[d], HIPFFT_Z2Z, (int)(n / lens[d]));
                hipfftSetStream(l, st);
                std::vector<double> hc2(2 * n);
                for (auto &x : hc2) x = rand() / (double)RAND_MAX - 0.5;
                hipMemcpyAsync(dstor, hc2.data(), n * 16, hipMemcpyHostToDevice, st);
                hipfftExecZ2Z(l, dstor, dstor, HIPFFT_FORWARD);
                hipfftExecZ2Z(l, dstor, dstor, HIPFFT_BACKWARD);
                std::vector<double> back(2 * n);
                hipStreamSynchronize(st);
                hipMemcpy(back.data(), dstor, n * 16, hipMemcpyDeviceToHost);
                double e1 = 0;
                for (size_t q = 0; q < 2 * n; ++q) e1 = fmax(e1, fabs(back[q] / lens[d] - hc2[q]));
                printf("    1-D Z2Z len %d batch %d err %.2e\n", lens[d], (int)(n / lens[d]), e1);
                if (destroy) hipfftDestroy(l);
            }
        }
        if (destroy) { if (with_c2c) hipfftDestroy(c); hipfftDestroy(f); hipfftDestroy(b); hipFree(din); hipFree(dc); hipFree(dout); hipFree(dstor); }
    }
    return 0;
}
