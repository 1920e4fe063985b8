// ocn_transpose.h -- TransposableField and the transposing FFT solver of PENCIL partitions (included by ocn_dist.h).
//
// Reference: src/DistributedComputations/transposable_field.jl:49-105 (zfield / yfield / xfield on the twin grids, :122-182; the two
// sub-communicators of MPI.Comm_split, :81-82), distributed_transpose.jl:25-95 (the eight pack / unpack kernels), :185-191 (Alltoallv! with
// equal counts on the sub-communicator) and distributed_fft_based_poisson_solver.jl:141-178 (solve!: z transform, z -> y, y transform,
// y -> x, x transform, divide, and back).
//
// Partition(Rx, Ry), rank = ix Ry + iy (ocn_dist_set_layout). Complex (double2) fields, x fastest, no halos:
//   zfield  (nx, ny, Nz)    z-local: nx = Nx / Rx, ny = Ny / Ry                       -- what the source term is written into
//   yfield  (nx, Ny, nz)    y-local: nz = Nz / Ry   (twin grid, ranks (Rx, 1, Ry))     -- zfield itself when Ry = 1
//   xfield  (Nx, nyx, nz)   x-local: nyx = Ny / Rx  (twin grid, ranks (1, Rx, Ry))     -- yfield itself when Rx = 1
// z <-> y exchanges inside the group of the Ry ranks that share ix; y <-> x inside the group of the Rx ranks that share iy. The groups
// need no communicator of their own here: RCCL send / recv pairs name their peers (one group call per transpose -- on a pencil partition
// of one node every peer sits on its own xGMI link), a caller-supplied transport gets the list of peers (all_to_all_group).
#pragma once

// ---- pack / unpack: the reference's index formulas, 0-based; one thread per element of the field that is read (pack) / written (unpack)
__global__ void __launch_bounds__(256) pack_z_to_y_kernel(double2 *send, const double2 *zf, int nx, int ny, int Nz) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nx * ny * Nz) return;
    const int i = t % nx, j = (t / nx) % ny, k = (int)(t / ((long)nx * ny));
    send[j + (long)ny * (i + (long)nx * k)] = zf[t];                                   // yzbuff.send[j + Ny (i-1 + Nx (k-1))] (:25-29)
}
__global__ void __launch_bounds__(256) unpack_y_from_z_kernel(const double2 *recv, double2 *yf, int nx, int Ny, int nz, int ny) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nx * Ny * nz) return;
    const int i = t % nx, j = (t / nx) % Ny, k = (int)(t / ((long)nx * Ny));
    const int jp = j % ny, m = j / ny;                                                 // (:75-84) size = (N[1], n[2], N[3])
    yf[t] = recv[jp + (long)ny * (i + (long)nx * k) + (long)m * nx * ny * nz];
}
__global__ void __launch_bounds__(256) pack_y_to_x_kernel(double2 *send, const double2 *yf, int nx, int Ny, int nz) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nx * Ny * nz) return;
    const int i = t % nx, j = (t / nx) % Ny, k = (int)(t / ((long)nx * Ny));
    send[i + (long)nx * (k + (long)nz * j)] = yf[t];                                   // xybuff.send[i + Nx (k-1 + Nz (j-1))] (:38-42)
}
__global__ void __launch_bounds__(256) unpack_x_from_y_kernel(const double2 *recv, double2 *xf, int Nx, int nyx, int nz, int nx) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Nx * nyx * nz) return;
    const int i = t % Nx, j = (t / Nx) % nyx, k = (int)(t / ((long)Nx * nyx));
    const int ip = i % nx, m = i / nx;                                                 // (:51-60) size = (n[1], N[2], N[3])
    xf[t] = recv[ip + (long)nx * (k + (long)nz * j) + (long)m * nx * nyx * nz];
}
__global__ void __launch_bounds__(256) pack_x_to_y_kernel(double2 *send, const double2 *xf, int Nx, int nyx, int nz) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Nx * nyx * nz) return;
    const int i = t % Nx, j = (t / Nx) % nyx, k = (int)(t / ((long)Nx * nyx));
    send[j + (long)nyx * (k + (long)nz * i)] = xf[t];                                  // xybuff.send[j + Ny (k-1 + Nz (i-1))] (:31-35)
}
__global__ void __launch_bounds__(256) unpack_y_from_x_kernel(const double2 *recv, double2 *yf, int nx, int Ny, int nz, int nyx) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nx * Ny * nz) return;
    const int i = t % nx, j = (t / nx) % Ny, k = (int)(t / ((long)nx * Ny));
    const int jp = j % nyx, m = j / nyx;                                               // (:87-95) size = (N[1], n[2], N[3])
    yf[t] = recv[jp + (long)nyx * (k + (long)nz * i) + (long)m * nx * nyx * nz];
}
__global__ void __launch_bounds__(256) pack_y_to_z_kernel(double2 *send, const double2 *yf, int nx, int Ny, int nz) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nx * Ny * nz) return;
    const int i = t % nx, j = (t / nx) % Ny, k = (int)(t / ((long)nx * Ny));
    send[k + (long)nz * (i + (long)nx * j)] = yf[t];                                   // xybuff.send[k + Nz (i-1 + Nx (j-1))] (:45-49)
}
__global__ void __launch_bounds__(256) unpack_z_from_y_kernel(const double2 *recv, double2 *zf, int nx, int ny, int Nz, int nz) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nx * ny * Nz) return;
    const int i = t % nx, j = (t / nx) % ny, k = (int)(t / ((long)nx * ny));
    const int kp = k % nz, m = k / nz;                                                 // (:63-72) size = (N[1], N[2], n[3])
    zf[t] = recv[kp + (long)nz * (i + (long)nx * j) + (long)m * nx * ny * nz];
}

struct ocn_transposable_s {
    ocn_dist_t dist = nullptr;
    int Rx = 1, Ry = 1, ix = 0, iy = 0;
    int Nx = 0, Ny = 0, Nz = 0, nx = 0, ny = 0, nz = 0, nyx = 0;
    double2 *zfield = nullptr, *yfield = nullptr, *xfield = nullptr;
    double2 *yz_send = nullptr, *yz_recv = nullptr, *xy_send = nullptr, *xy_recv = nullptr;
    int yz_peers[OCN_MAX_RANKS], xy_peers[OCN_MAX_RANKS];       // world ranks of the two groups, in group order
    long local() const { return (long)nx * ny * Nz; }            // elements of every one of the three fields
};

// MPI.Alltoallv! with equal counts on a sub-communicator (distributed_transpose.jl:185-191): chunk g of `send` goes to peers[g], chunk g
// of `recv` comes from peers[g]; `count` doubles per chunk; ordered on the compute stream
static int dist_all_to_all_group(ocn_dist_t d, const int *peers, int npeers, const double *send, double *recv, size_t count) {
    if (d->kind == 1) {
        if (!d->tr.all_to_all_group) return fail(OCN_ENOTSUP, "this transport has no all_to_all_group entry (needed by the pencil transposes)");
        int rc = d->tr.all_to_all_group(d->tr.user, peers, npeers, send, recv, count, (void *)g_stream);
        return rc ? fail(rc, "transport all_to_all_group failed") : OCN_OK;
    }
    NCCL_TRY(g_rccl.GroupStart());
    for (int q = 0; q < npeers; ++q) {
        NCCL_TRY_IN_GROUP(g_rccl.Send(send + (size_t)q * count, count, OCN_NCCL_FLOAT64, peers[q], d->comm, g_stream));
        NCCL_TRY_IN_GROUP(g_rccl.Recv(recv + (size_t)q * count, count, OCN_NCCL_FLOAT64, peers[q], d->comm, g_stream));
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return OCN_OK;
}

extern "C" int ocn_transposable_destroy(ocn_transposable_t t) {
    if (!t) return OCN_OK;
    if (t->yfield != t->zfield) hipFree(t->yfield);
    if (t->xfield != t->yfield) hipFree(t->xfield);
    hipFree(t->zfield);
    hipFree(t->yz_send); hipFree(t->yz_recv); hipFree(t->xy_send); hipFree(t->xy_recv);
    delete t;
    return OCN_OK;
}

// TransposableField(field_in, ComplexF64) for a field of GLOBAL size (Nx, Ny, Nz) on the communicator's Partition(Rx, Ry)
// (transposable_field.jl:49-105). The reference's transposes move equal chunks: Rx | Nx, Ry | Ny, Ry | Nz, Rx | Ny
// (distributed_fft_based_poisson_solver.jl:213-226).
extern "C" int ocn_transposable_create(ocn_transposable_t *out, ocn_dist_t dist, int Nx, int Ny, int Nz) {
    NEED_INIT();
    if (!out || !dist || Nx < 1 || Ny < 1 || Nz < 1) return fail(OCN_EINVAL, "invalid argument");
    const int Rx = dist->Rx, Ry = dist->Ry;
    if (dist->world > OCN_MAX_RANKS) return fail(OCN_ENOTSUP, "at most %d ranks", OCN_MAX_RANKS);
    if (Nx % Rx || Ny % Ry) return fail(OCN_EINVAL, "the partition (%d, %d) must divide the horizontal size (%d, %d)", Rx, Ry, Nx, Ny);
    if (Ry > 1 && Nz % Ry) return fail(OCN_EINVAL, "Nz = %d must be divisible by Ry = %d (transpose z -> y)", Nz, Ry);
    if (Rx > 1 && Ny % Rx) return fail(OCN_EINVAL, "Ny = %d must be divisible by Rx = %d (transpose y -> x)", Ny, Rx);
    ocn_transposable_s *t = new ocn_transposable_s();
    t->dist = dist; t->Rx = Rx; t->Ry = Ry; t->ix = dist->rank / Ry; t->iy = dist->rank % Ry;
    t->Nx = Nx; t->Ny = Ny; t->Nz = Nz; t->nx = Nx / Rx; t->ny = Ny / Ry; t->nz = Nz / Ry; t->nyx = Ny / Rx;
    for (int q = 0; q < Ry; ++q) t->yz_peers[q] = t->ix * Ry + q;          // same ix: MPI.Comm_split(COMM_WORLD, local_index[1], ...)
    for (int q = 0; q < Rx; ++q) t->xy_peers[q] = q * Ry + t->iy;          // same iy (the z index of the y-local twin architecture)
    const size_t bytes = (size_t)t->local() * sizeof(double2);
    auto alloc = [&](double2 **p) {
        hipError_t e = dev_alloc((void **)p, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, g_stream);
        return e;
    };
    hipError_t e = alloc(&t->zfield);
    t->yfield = t->zfield;
    if (e == hipSuccess && Ry > 1) { e = alloc(&t->yfield); if (e == hipSuccess) e = alloc(&t->yz_send); if (e == hipSuccess) e = alloc(&t->yz_recv); }
    t->xfield = t->yfield;
    if (e == hipSuccess && Rx > 1) { e = alloc(&t->xfield); if (e == hipSuccess) e = alloc(&t->xy_send); if (e == hipSuccess) e = alloc(&t->xy_recv); }
    if (e != hipSuccess) { ocn_transposable_destroy(t); return fail((int)e, "dev_alloc(TransposableField): %s", hipGetErrorString(e)); }
    *out = t;
    return OCN_OK;
}

// device pointers (complex, interleaved) and sizes of the three configurations
extern "C" int ocn_transposable_fields(ocn_transposable_t t, double **zfield, double **yfield, double **xfield, int zsize[3], int ysize[3], int xsize[3]) {
    if (!t) return fail(OCN_EINVAL, "NULL argument");
    if (zfield) *zfield = (double *)t->zfield;
    if (yfield) *yfield = (double *)t->yfield;
    if (xfield) *xfield = (double *)t->xfield;
    if (zsize) { zsize[0] = t->nx; zsize[1] = t->ny; zsize[2] = t->Nz; }
    if (ysize) { ysize[0] = t->nx; ysize[1] = t->Ry > 1 ? t->Ny : t->ny; ysize[2] = t->Ry > 1 ? t->nz : t->Nz; }
    if (xsize) { xsize[0] = t->Rx > 1 ? t->Nx : t->nx; xsize[1] = t->Rx > 1 ? t->nyx : (t->Ry > 1 ? t->Ny : t->ny); xsize[2] = t->Ry > 1 ? t->nz : t->Nz; }
    return OCN_OK;
}

#define OCN_TRANSPOSE_LAUNCH(kernel, n, ...)                                                                          \
    do {                                                                                                                \
        hipLaunchKernelGGL(kernel, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, g_stream, __VA_ARGS__);           \
        KERNEL_CHECK();                                                                                                 \
    } while (0)

// transpose_z_to_y! / _y_to_x! / _x_to_y! / _y_to_z! (distributed_transpose.jl:185-191): pack, all-to-all inside the group, unpack;
// no-ops on slab partitions (:12-15)
extern "C" int ocn_transpose_z_to_y(ocn_transposable_t t) {
    NEED_INIT();
    if (!t) return fail(OCN_EINVAL, "NULL argument");
    if (t->Ry == 1) return OCN_OK;
    const long n = t->local();
    OCN_TRANSPOSE_LAUNCH(pack_z_to_y_kernel, n, t->yz_send, (const double2 *)t->zfield, t->nx, t->ny, t->Nz);
    int rc = dist_all_to_all_group(t->dist, t->yz_peers, t->Ry, (const double *)t->yz_send, (double *)t->yz_recv, 2 * (size_t)(n / t->Ry));
    if (rc) return rc;
    OCN_TRANSPOSE_LAUNCH(unpack_y_from_z_kernel, n, (const double2 *)t->yz_recv, t->yfield, t->nx, t->Ny, t->nz, t->ny);
    return OCN_OK;
}
extern "C" int ocn_transpose_y_to_z(ocn_transposable_t t) {
    NEED_INIT();
    if (!t) return fail(OCN_EINVAL, "NULL argument");
    if (t->Ry == 1) return OCN_OK;
    const long n = t->local();
    OCN_TRANSPOSE_LAUNCH(pack_y_to_z_kernel, n, t->yz_send, (const double2 *)t->yfield, t->nx, t->Ny, t->nz);
    int rc = dist_all_to_all_group(t->dist, t->yz_peers, t->Ry, (const double *)t->yz_send, (double *)t->yz_recv, 2 * (size_t)(n / t->Ry));
    if (rc) return rc;
    OCN_TRANSPOSE_LAUNCH(unpack_z_from_y_kernel, n, (const double2 *)t->yz_recv, t->zfield, t->nx, t->ny, t->Nz, t->nz);
    return OCN_OK;
}
extern "C" int ocn_transpose_y_to_x(ocn_transposable_t t) {
    NEED_INIT();
    if (!t) return fail(OCN_EINVAL, "NULL argument");
    if (t->Rx == 1) return OCN_OK;
    const long n = t->local();
    const int Nyl = t->Ry > 1 ? t->Ny : t->ny, nzl = t->Ry > 1 ? t->nz : t->Nz;        // the y-local configuration (zfield itself on x-slabs)
    OCN_TRANSPOSE_LAUNCH(pack_y_to_x_kernel, n, t->xy_send, (const double2 *)t->yfield, t->nx, Nyl, nzl);
    int rc = dist_all_to_all_group(t->dist, t->xy_peers, t->Rx, (const double *)t->xy_send, (double *)t->xy_recv, 2 * (size_t)(n / t->Rx));
    if (rc) return rc;
    OCN_TRANSPOSE_LAUNCH(unpack_x_from_y_kernel, n, (const double2 *)t->xy_recv, t->xfield, t->Nx, Nyl / t->Rx, nzl, t->nx);
    return OCN_OK;
}
extern "C" int ocn_transpose_x_to_y(ocn_transposable_t t) {
    NEED_INIT();
    if (!t) return fail(OCN_EINVAL, "NULL argument");
    if (t->Rx == 1) return OCN_OK;
    const long n = t->local();
    const int Nyl = t->Ry > 1 ? t->Ny : t->ny, nzl = t->Ry > 1 ? t->nz : t->Nz;
    OCN_TRANSPOSE_LAUNCH(pack_x_to_y_kernel, n, t->xy_send, (const double2 *)t->xfield, t->Nx, Nyl / t->Rx, nzl);
    int rc = dist_all_to_all_group(t->dist, t->xy_peers, t->Rx, (const double *)t->xy_send, (double *)t->xy_recv, 2 * (size_t)(n / t->Rx));
    if (rc) return rc;
    OCN_TRANSPOSE_LAUNCH(unpack_y_from_x_kernel, n, (const double2 *)t->xy_recv, t->yfield, t->nx, Nyl, nzl, Nyl / t->Rx);
    return OCN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// DistributedFFTBasedPoissonSolver on a pencil partition (distributed_fft_based_poisson_solver.jl:92-188): triply Periodic global grid,
// complex transforms one direction at a time, each on the configuration in which that direction is local.
// ---------------------------------------------------------------------------------------------------------------------
// `@. ϕc = -b / (λx + λy + λz)` and ϕc[1, 1, 1] = 0 (:163-168) on the x-local configuration; the 1 / (Nx Ny Nz) of the three inverse
// transforms rides along
__global__ void __launch_bounds__(256) pencil_divide_kernel(double2 *xf, const double *lx, const double *ly, const double *lz, int Nx, int nyx, int nz,
                                                            int j0, int k0, double scale) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Nx * nyx * nz) return;
    const int i = t % Nx, j = (t / Nx) % nyx, k = (int)(t / ((long)Nx * nyx));
    const double lam = (lx[i] + ly[j0 + j]) + lz[k0 + k] - 0.0;
    double2 v = xf[t];
    v.x = -v.x / lam * scale; v.y = -v.y / lam * scale;
    if (i == 0 && j0 + j == 0 && k0 + k == 0) v = make_double2(0.0, 0.0);
    xf[t] = v;
}

struct PencilSolve {
    ocn_transposable_s *tf = nullptr;
    hipfftHandle plan_z = 0, plan_y = 0, plan_x = 0;
    bool has_z = false, has_y = false, has_x = false;
    double *lam[3] = {nullptr, nullptr, nullptr};
    int ny_planes = 0;             // the y transform runs plane by plane (lines with stride nx inside a (nx, Ny) plane)
};

static void pencil_solve_free(PencilSolve *p) {
    if (!p) return;
    if (p->has_z) hipfftDestroy(p->plan_z);
    if (p->has_y) hipfftDestroy(p->plan_y);
    if (p->has_x) hipfftDestroy(p->plan_x);
    for (int d = 0; d < 3; ++d) hipFree(p->lam[d]);
    ocn_transposable_destroy(p->tf);
    delete p;
}

static int pencil_solve_create(PencilSolve **out, ocn_dist_t dist, const int N[3], const double L[3]) {
    PencilSolve *p = new PencilSolve();
    int rc = ocn_transposable_create(&p->tf, dist, N[0], N[1], N[2]);
    if (rc) { delete p; return rc; }
    ocn_transposable_s *t = p->tf;
    for (int d = 0; d < 3; ++d) {
        std::vector<double> lam;
        poisson_eigenvalues(N[d], L[d], OCN_PERIODIC, lam);
        hipError_t e = dev_alloc((void **)&p->lam[d], N[d] * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(p->lam[d], lam.data(), N[d] * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { pencil_solve_free(p); return fail((int)e, "pencil solver eigenvalues: %s", hipGetErrorString(e)); }
    }
    const int Nyl = t->Ry > 1 ? t->Ny : t->ny, nzl = t->Ry > 1 ? t->nz : t->Nz;
    const int Nxl = t->Rx > 1 ? t->Nx : t->nx, nyxl = t->Rx > 1 ? Nyl / t->Rx : Nyl;
    {   // z lines of zfield (nx, ny, Nz): stride nx ny, one line per (i, j)
        int n[1] = {t->Nz};
        const int plane = t->nx * t->ny;
        hipfftResult r = hipfftPlanMany(&p->plan_z, 1, n, n, plane, 1, n, plane, 1, HIPFFT_Z2Z, plane);
        if (r != HIPFFT_SUCCESS) { pencil_solve_free(p); return fail(1000 + (int)r, "hipfftPlanMany(pencil z) failed (%d)", (int)r); }
        p->has_z = true;
    }
    {   // y lines of yfield (nx, Ny, nz): stride nx inside one (nx, Ny) plane, one plane per exec
        int n[1] = {Nyl};
        hipfftResult r = hipfftPlanMany(&p->plan_y, 1, n, n, t->nx, 1, n, t->nx, 1, HIPFFT_Z2Z, t->nx);
        if (r != HIPFFT_SUCCESS) { pencil_solve_free(p); return fail(1000 + (int)r, "hipfftPlanMany(pencil y) failed (%d)", (int)r); }
        p->has_y = true;
        p->ny_planes = nzl;
    }
    {   // x lines of xfield (Nx, nyx, nz): unit stride
        int n[1] = {Nxl};
        hipfftResult r = hipfftPlanMany(&p->plan_x, 1, n, nullptr, 1, Nxl, nullptr, 1, Nxl, HIPFFT_Z2Z, nyxl * nzl);
        if (r != HIPFFT_SUCCESS) { pencil_solve_free(p); return fail(1000 + (int)r, "hipfftPlanMany(pencil x) failed (%d)", (int)r); }
        p->has_x = true;
    }
    if ((rc = plan_set_stream(p->plan_z)) || (rc = plan_set_stream(p->plan_y)) || (rc = plan_set_stream(p->plan_x))) { pencil_solve_free(p); return rc; }
    // every plan is checked by a pseudo-random round trip before it is trusted
    if ((rc = verify_complex_plan(p->plan_z, t->zfield, t->local(), 1.0 / (double)t->Nz, "pencil z")) ||
        (rc = verify_complex_plan(p->plan_y, t->yfield, (long)t->nx * Nyl, 1.0 / (double)Nyl, "pencil y")) ||
        (rc = verify_complex_plan(p->plan_x, t->xfield, t->local(), 1.0 / (double)Nxl, "pencil x"))) { pencil_solve_free(p); return rc; }
    hipMemsetAsync(t->zfield, 0, (size_t)t->local() * sizeof(double2), g_stream);
    if (t->yfield != t->zfield) hipMemsetAsync(t->yfield, 0, (size_t)t->local() * sizeof(double2), g_stream);
    if (t->xfield != t->yfield) hipMemsetAsync(t->xfield, 0, (size_t)t->local() * sizeof(double2), g_stream);
    *out = p;
    return OCN_OK;
}

// solve!(x, solver::DistributedFFTBasedPoissonSolver) (:141-178); the right-hand side is in tf->zfield, the solution returns there
static int pencil_solve(PencilSolve *p) {
    ocn_transposable_s *t = p->tf;
    const int Nyl = t->Ry > 1 ? t->Ny : t->ny, nzl = t->Ry > 1 ? t->nz : t->Nz;
    const int Nxl = t->Rx > 1 ? t->Nx : t->nx, nyxl = t->Rx > 1 ? Nyl / t->Rx : Nyl;
    auto ydir = [&](int dir) -> int {
        for (int k = 0; k < p->ny_planes; ++k) {
            hipfftDoubleComplex *pl = (hipfftDoubleComplex *)(t->yfield + (long)k * t->nx * Nyl);
            FFT_TRY(hipfftExecZ2Z(p->plan_y, pl, pl, dir));
        }
        return OCN_OK;
    };
    int rc;
    FFT_TRY(hipfftExecZ2Z(p->plan_z, (hipfftDoubleComplex *)t->zfield, (hipfftDoubleComplex *)t->zfield, HIPFFT_FORWARD));
    if ((rc = ocn_transpose_z_to_y(t))) return rc;
    if ((rc = ydir(HIPFFT_FORWARD))) return rc;
    if ((rc = ocn_transpose_y_to_x(t))) return rc;
    FFT_TRY(hipfftExecZ2Z(p->plan_x, (hipfftDoubleComplex *)t->xfield, (hipfftDoubleComplex *)t->xfield, HIPFFT_FORWARD));
    // x-local twin architecture: ranks (1, Rx, Ry) -- this rank's y block is its ix, its z block its iy (transposable_field.jl:160-170)
    const int j0 = t->Rx > 1 ? t->ix * nyxl : (t->Ry > 1 ? 0 : t->iy * t->ny), k0 = t->Ry > 1 ? t->iy * nzl : 0;
    const long n = t->local();
    const double scale = 1.0 / ((double)t->Nx * (double)t->Ny * (double)t->Nz);
    hipLaunchKernelGGL(pencil_divide_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, t->xfield, p->lam[0], p->lam[1], p->lam[2], Nxl,
                       nyxl, nzl, j0, k0, scale);
    KERNEL_CHECK();
    FFT_TRY(hipfftExecZ2Z(p->plan_x, (hipfftDoubleComplex *)t->xfield, (hipfftDoubleComplex *)t->xfield, HIPFFT_BACKWARD));
    if ((rc = ocn_transpose_x_to_y(t))) return rc;
    if ((rc = ydir(HIPFFT_BACKWARD))) return rc;
    if ((rc = ocn_transpose_y_to_z(t))) return rc;
    FFT_TRY(hipfftExecZ2Z(p->plan_z, (hipfftDoubleComplex *)t->zfield, (hipfftDoubleComplex *)t->zfield, HIPFFT_BACKWARD));
    return OCN_OK;
}
