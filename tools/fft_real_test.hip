// standalone exploration: hipFFT 3-D real transforms round trip, several plan flavours and sizes
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
int main() {
    int sizes[][3] = {{32,16,8},{16,16,16},{8,16,32},{32,32,32},{64,16,8},{32,16,16},{64,64,64},{256,256,256},{12,10,6},{16,12,10}};
    for (auto &sz : sizes) {
        int Nx = sz[0], Ny = sz[1], Nz = sz[2], Nxh = Nx / 2 + 1;
        size_t n = (size_t)Nx * Ny * Nz, nh = (size_t)Nxh * Ny * Nz;
        std::vector<double> h(n), out(n);
        srand(1);
        for (auto &x : h) x = rand() / (double)RAND_MAX - 0.5;
        for (int flavour = 0; flavour < 3; ++flavour) {
            double *din, *dout; hipfftDoubleComplex *dc;
            int H = 3, Px = Nx + 2 * H, Py = Ny + 2 * H, Pz = Nz + 2 * H;
            size_t np = (size_t)Px * Py * Pz;
            hipMalloc(&din, n * 8); hipMalloc(&dc, nh * 16); hipMalloc(&dout, (flavour == 2 ? np : n) * 8);
            hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice);
            hipMemset(dout, 0, (flavour == 2 ? np : n) * 8);
            hipfftHandle f, b;
            int n3[3] = {Nz, Ny, Nx};
            hipfftResult r1, r2;
            if (flavour == 0) { r1 = hipfftPlan3d(&f, Nz, Ny, Nx, HIPFFT_D2Z); r2 = hipfftPlan3d(&b, Nz, Ny, Nx, HIPFFT_Z2D); }
            else if (flavour == 1) { r1 = hipfftPlanMany(&f, 3, n3, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, 1); r2 = hipfftPlanMany(&b, 3, n3, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, 1); }
            else { int ie[3] = {Nz, Ny, Nxh}, oe[3] = {Pz, Py, Px};
                   r1 = hipfftPlanMany(&f, 3, n3, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, 1);
                   r2 = hipfftPlanMany(&b, 3, n3, ie, 1, (int)nh, oe, 1, (int)np, HIPFFT_Z2D, 1); }
            if (r1 || r2) { printf("%dx%dx%d flavour %d plan fail %d %d\n", Nx, Ny, Nz, flavour, r1, r2); continue; }
            hipfftExecD2Z(f, din, dc);
            double *o = flavour == 2 ? dout + H + (size_t)Px * (H + (size_t)Py * H) : dout;
            hipfftExecZ2D(b, dc, o);
            hipDeviceSynchronize();
            double err = 0;
            if (flavour == 2) {
                std::vector<double> p(np); hipMemcpy(p.data(), dout, np * 8, hipMemcpyDeviceToHost);
                double halo = 0;
                for (int k = 0; k < Pz; ++k) for (int j = 0; j < Py; ++j) for (int i = 0; i < Px; ++i) {
                    double v = p[i + (size_t)Px * (j + (size_t)Py * k)];
                    bool in = i >= H && i < H + Nx && j >= H && j < H + Ny && k >= H && k < H + Nz;
                    if (in) err = fmax(err, fabs(v / n - h[(i - H) + (size_t)Nx * ((j - H) + (size_t)Ny * (k - H))]));
                    else halo = fmax(halo, fabs(v));
                }
                printf("%dx%dx%d strided-into-halo roundtrip err %.2e  halo-touched %.2e\n", Nx, Ny, Nz, err, halo);
            } else {
                hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost);
                for (size_t q = 0; q < n; ++q) err = fmax(err, fabs(out[q] / n - h[q]));
                printf("%dx%dx%d %s roundtrip err %.2e\n", Nx, Ny, Nz, flavour == 0 ? "Plan3d" : "PlanMany-dense", err);
            }
            hipfftDestroy(f); hipfftDestroy(b); hipFree(din); hipFree(dc); hipFree(dout);
        }
    }
    return 0;
}
