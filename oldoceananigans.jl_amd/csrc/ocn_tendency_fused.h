// ocn_tendency_fused.h -- fused, flux-sharing WENO-5 tendency kernel: Gu, Gv, Gw and all tracer tendencies in ONE pass.
//
// Reference formulation (compute_nonhydrostatic_tendencies.jl:49-163): 3 + ntracers launches, every thread evaluates
// the fluxes on BOTH faces of its cell in all three directions, i.e. each face flux is computed twice, and u, v, w
// are re-read by every launch (22 array passes for 2 tracers).
//
// MI355X formulation (this file): every face flux is evaluated exactly ONCE and handed to the neighbouring cell.
//   * a workgroup owns a (64 x TY) column tile and MARCHES along z; each thread keeps the 6-deep z-windows of all
//     prognostic fields in registers (coalesced 512-B row loads, one new plane per iteration), so the z-fluxes need no
//     neighbour at all: the flux through the bottom face of cell k+1 is the top flux of cell k one iteration later;
//   * each thread evaluates only the LOW-side x-, y-, z-fluxes of its cell for every field; the high-side x-flux comes
//     from lane+1 through a wavefront shuffle, the high-side y-flux from the next row (next wave) through LDS
//     (double-buffered, one s_barrier per plane);
//   * one extra wave per workgroup evaluates the y-fluxes of the row just above the tile and the x-fluxes of the
//     column just right of it, so tiles do not overlap and nothing is recomputed except those edge faces
//     (1/TY of the y-fluxes + 1/64 of the x-fluxes);
//   * x / y stencil neighbours are read straight from global memory (L1/L2 hits: the kernel is FP64-issue bound, not
//     bandwidth bound -- see DESIGN.md roofline discussion).
// The arithmetic of every flux is the same IEEE sequence as the per-field kernels => results are bit-identical.
#pragma once
#include "ocn_device.h"

#define OCN_FUSED_MAXTR 8

struct FusedArgs {
    const double *u, *v, *w;
    const double *c[OCN_FUSED_MAXTR];
    double *Gu, *Gv, *Gw;
    double *Gc[OCN_FUSED_MAXTR];
    int s1;            // common x-row stride of all fields (x, y periodic => identical parent x/y extents)
    long s2u, s2w;     // plane strides: (u, v, tracers) share s2u; w may have one more plane on Bounded z
    long off;          // (Hx-1) + s1 (Hy-1): offset of 1-based (i, j) inside a plane; planes are indexed k-1+Hz
    Range6 r;          // cell range of the launch
    Range6 ru, rv, rw, rc;  // per-field store masks (exclude_periphery)
    int kchunk;        // levels per workgroup along z
    // optional fused RK3 substep of the NEXT stage (runge_kutta_3.jl:212-226): Un = U + dt (gamma Gn + zeta Gm), written to a
    // second set of prognostic arrays (other workgroups still read U). substep = 0: tendencies only.
    int substep, has_zeta;
    // 1: workgroups that share an XCD (linear id mod 8 under the observed round-robin dispatch) get a CONTIGUOUS range of tiles, so
    // the halo columns / rows of neighbouring tiles are found in that XCD's own L2 (bijective remap; a pure speed choice)
    int xcd_swizzle;
    double dt, gamma, zeta;
    double *Un[3 + OCN_FUSED_MAXTR];
    const double *Gm[3 + OCN_FUSED_MAXTR];
};

// window loaders ----------------------------------------------------------------------------------------------------
struct Win6 { double s[6]; };

// Addressing: every field shares one parent layout, so a thread carries ONE 32-bit byte offset per plane and each access is
// `uniform base (SGPR pair) + 32-bit VGPR offset (+ immediate)` -- the gfx950 global_load saddr form; no 64-bit per-lane
// address arithmetic (it was 16 % of the VALU stream). Parent arrays are < 4 GiB (checked on the host).
__device__ __forceinline__ double ld8(const double *base, unsigned byte_off) {
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void st8(double *base, unsigned byte_off, double v) {
    *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
__device__ __forceinline__ Win6 load_win(const double *base, unsigned o, unsigned stride) {
    Win6 w;
#pragma unroll
    for (int n = 0; n < 6; ++n) w.s[n] = ld8(base, o + (unsigned)(n - 3) * stride);
    return w;
}

template <int ARITH = 0>
__device__ __forceinline__ double sym4(const Win6 &q, double a, bool bounded, int idx, bool center, int N) {
    // ARITH 1 (ocn_device.h): one multiplication by the uniform area after the interpolation instead of four before it
    if (ARITH == 1) return a * symmetric_interp(q.s[1], q.s[2], q.s[3], q.s[4], bounded, idx, center, N);
    return symmetric_interp(a * q.s[1], a * q.s[2], a * q.s[3], a * q.s[4], bounded, idx, center, N);
}
// symmetric interpolation along z: per-level area factors a[k-2 .. k+1]
__device__ __forceinline__ double sym4z(const Win6 &q, const double *a, bool bounded, int idx, bool center, int N) {
    return symmetric_interp(a[0] * q.s[1], a[1] * q.s[2], a[2] * q.s[3], a[3] * q.s[4], bounded, idx, center, N);
}
template <int ARITH = 0>
__device__ __forceinline__ double bias6(const Win6 &s, bool left, bool bounded, int idx, bool center, int N) {
    return biased_interp<ARITH>(s.s[0], s.s[1], s.s[2], s.s[3], s.s[4], s.s[5], left, bounded, idx, center, N);
}

// The low-side fluxes of cell (i, j, k) (reference: upwind_biased_advective_fluxes.jl:23-121):
//   x: Uu(i-1), Uv(i), Uw(i), cx(i);   y: Vu(j), Vv(j-1), Vw(j), cy(j);   z: Wu(k), Wv(k), Ww(k-1), cz(k)
template <int NTR>
__device__ __forceinline__ void x_fluxes(const DGrid &g, int i, int j, int k, double axk, const double *axz,
                                         const Win6 &ux, const Win6 &uy, const Win6 &uz, const Win6 &vx, const Win6 &wx,
                                         const Win6 *cx, double *F) {
    const bool bx = g.tx != 0, by = g.ty != 0, bz = g.tz != 0;
    double ut = sym4(ux, axk, bx, i - 1, true, g.Nx);                       // advective_momentum_flux_Uu :23-29
    F[0] = ut * bias6(ux, ut > 0, bx, i - 1, true, g.Nx);
    ut = sym4(uy, axk, by, j, false, g.Ny);                                 // Uv :47-53
    F[1] = ut * bias6(vx, ut > 0, bx, i, false, g.Nx);
    ut = sym4z(uz, axz, bz, k, false, g.Nz);                                // Uw :71-77
    F[2] = ut * bias6(wx, ut > 0, bx, i, false, g.Nx);
    const double u0 = ux.s[3];
#pragma unroll
    for (int t = 0; t < NTR; ++t) F[3 + t] = axk * u0 * bias6(cx[t], u0 > 0, bx, i, false, g.Nx);   // :99-105
}

template <int NTR>
__device__ __forceinline__ void y_fluxes(const DGrid &g, int i, int j, int k, double ayk, const double *ayz,
                                         const Win6 &vx, const Win6 &vy, const Win6 &vz, const Win6 &uy, const Win6 &wy,
                                         const Win6 *cy, double *F) {
    const bool bx = g.tx != 0, by = g.ty != 0, bz = g.tz != 0;
    double vt = sym4(vx, ayk, bx, i, false, g.Nx);                          // Vu :31-37
    F[0] = vt * bias6(uy, vt > 0, by, j, false, g.Ny);
    vt = sym4(vy, ayk, by, j - 1, true, g.Ny);                              // Vv :55-61
    F[1] = vt * bias6(vy, vt > 0, by, j - 1, true, g.Ny);
    vt = sym4z(vz, ayz, bz, k, false, g.Nz);                                // Vw :79-85
    F[2] = vt * bias6(wy, vt > 0, by, j, false, g.Ny);
    const double v0 = vy.s[3];
#pragma unroll
    for (int t = 0; t < NTR; ++t) F[3 + t] = ayk * v0 * bias6(cy[t], v0 > 0, by, j, false, g.Ny);   // :107-113
}

template <int NTR>
__device__ __forceinline__ void z_fluxes(const DGrid &g, int i, int j, int k, const Win6 &wx, const Win6 &wy, const Win6 &wz,
                                         const Win6 &uz, const Win6 &vz, const Win6 *cz, double *F) {
    const bool bx = g.tx != 0, by = g.ty != 0, bz = g.tz != 0;
    const double az = g.az;
    double wt = sym4(wx, az, bx, i, false, g.Nx);                           // Wu :39-45
    F[0] = wt * bias6(uz, wt > 0, bz, k, false, g.Nz);
    wt = sym4(wy, az, by, j, false, g.Ny);                                  // Wv :63-69
    F[1] = wt * bias6(vz, wt > 0, bz, k, false, g.Nz);
    wt = sym4(wz, az, bz, k - 1, true, g.Nz);                               // Ww :87-93
    F[2] = wt * bias6(wz, wt > 0, bz, k - 1, true, g.Nz);
    const double w0 = wz.s[3];
#pragma unroll
    for (int t = 0; t < NTR; ++t) F[3 + t] = az * w0 * bias6(cz[t], w0 > 0, bz, k, false, g.Nz);    // :115-121
}

__device__ __forceinline__ bool in_range(const Range6 &r, int i, int j, int k) {
    return i >= r.i0 && i <= r.i1 && j >= r.j0 && j <= r.j1 && k >= r.k0 && k <= r.k1;
}

__device__ __forceinline__ double shfl_down1(double x) {
    return __shfl_down(x, 1, 64);
}

// One workgroup = (TY + 1) wavefronts of 64 lanes: waves 0..TY-1 own tile rows, wave TY is the edge wave.
// Register diet (v2): nothing but the previous plane's z-fluxes is carried across iterations; all low-side x / y fluxes of
// a plane go to LDS (double buffered) and are read back -- own and neighbour's -- when the cell is closed one iteration
// later, so the kernel fits 4+ waves per SIMD. BZ = z is Bounded (compile-time: the periodic build carries no fallback
// reconstruction code).
template <int NTR, int TY, bool BZ, int MINW, bool ZWIN>
__global__ void __launch_bounds__(64 * (TY + 1), MINW) fused_tendency_kernel(DGrid gin, FusedArgs a) {
    constexpr int NF = 3 + NTR;
    constexpr int NA = NTR > 0 ? NTR : 1;   // no zero-length arrays in device code
    __shared__ double FX[2][NF][TY][66];            // low-side x-fluxes of columns 0..64 (65 used, padded)
    __shared__ double FY[2][NF][TY + 1][64];        // low-side y-fluxes of rows 0..TY

    DGrid g = gin;
    g.tx = 0; g.ty = 0; g.tz = BZ ? 1 : 0;          // compile-time topology (x, y periodic is a launch precondition)

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd_swizzle) {
        const unsigned gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
        const unsigned orig = bx + gx * (by + gy * bz);
        const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
        bx = id % gx; by = (id / gx) % gy; bz = id / (gx * gy);
    }
    const int i0 = a.r.i0 + bx * 64, j0 = a.r.j0 + by * TY;
    const int kc0 = a.r.k0 + bz * a.kchunk;
    const int kc1 = min(kc0 + a.kchunk - 1, a.r.k1);
    const bool edge = wave == TY;
    const int i = i0 + lane;
    const int j = j0 + wave;                          // edge wave: row j0 + TY
    // cells whose tendencies this thread produces / faces whose low fluxes it must evaluate
    const bool cell_ij = !edge && i <= a.r.i1 && j <= a.r.j1;
    const bool flux_ij = !edge && i <= a.r.i1 + 1 && j <= a.r.j1 + 1;
    // edge wave roles: the row above the tile (all lanes) and the column right of it (one lane per row)
    const bool edge_y = edge && i <= a.r.i1 && j <= a.r.j1 + 1;
    const int ie = i0 + 64, je = j0 + lane;
    const bool edge_x = edge && lane < TY && ie <= a.r.i1 + 1 && je <= a.r.j1;

    const unsigned s1 = 8u * (unsigned)a.s1, s2 = 8u * (unsigned)a.s2u, sx = 8u;     // byte strides
    const int Hz = g.Hz;
    const unsigned col = 8u * (unsigned)(a.off + i + (long)a.s1 * j);       // (i, j) of this thread inside a plane (edge wave: row j0+TY)
    const unsigned cole = 8u * (unsigned)(a.off + ie + (long)a.s1 * je);

    double fz_prev[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) fz_prev[f] = 0;

    // ZWIN: the 6-deep z-windows of the thread's own column live in registers and slide by one plane per iteration, so
    // every plane of every field is fetched from HBM once per tile instead of six times (the L2 cannot hold the windows
    // of all co-resident workgroups: 5 fields x 6 planes x tile ~ 0.2 MB per workgroup)
    Win6 uz, vz, wz, czw[NA];
    if (ZWIN && flux_ij) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const unsigned oz = col + s2 * (unsigned)(kc0 - 3 + n - 1 + Hz);
            uz.s[n + 1] = ld8(a.u, oz);
            vz.s[n + 1] = ld8(a.v, oz);
            wz.s[n + 1] = ld8(a.w, oz);
#pragma unroll
            for (int t = 0; t < NTR; ++t) czw[t].s[n + 1] = ld8(a.c[t], oz);
        }
    }

    for (int k = kc0; k <= kc1 + 1; ++k) {
        const bool last = k == kc1 + 1;               // peeled plane: only the z-fluxes closing cell kc1
        const int buf = k & 1;
        const long pk = (long)(k - 1 + Hz);
        const double axk = g.ax[pk], ayk = g.ay[pk];
        if (flux_ij) {
            const unsigned o = col + s2 * (unsigned)pk;
            const double *pu = a.u, *pv = a.v, *pw = a.w;
            // previous-stage tendencies of the cell closed in this iteration: issued first, consumed after the z-fluxes
            double gm[NF];
            const bool close_cell = k > kc0 && cell_ij;
            if (a.substep && a.has_zeta && close_cell) {
#pragma unroll
                for (int f = 0; f < NF; ++f) gm[f] = ld8(a.Gm[f], o - s2);
            }
            // ---- z-fluxes of plane k, then close cell k-1 ----
            double fz[NF];
            if (ZWIN) {
#pragma unroll
                for (int n = 0; n < 5; ++n) {
                    uz.s[n] = uz.s[n + 1]; vz.s[n] = vz.s[n + 1]; wz.s[n] = wz.s[n + 1];
#pragma unroll
                    for (int t = 0; t < NTR; ++t) czw[t].s[n] = czw[t].s[n + 1];
                }
                const unsigned ot = o + 2u * s2;
                uz.s[5] = ld8(a.u, ot);
                vz.s[5] = ld8(a.v, ot);
                wz.s[5] = ld8(a.w, ot);
#pragma unroll
                for (int t = 0; t < NTR; ++t) czw[t].s[5] = ld8(a.c[t], ot);
                z_fluxes<NTR>(g, i, j, k, load_win(pw, o, sx), load_win(pw, o, s1), wz, uz, vz, czw, fz);
            } else {
                Win6 cz[NA];
#pragma unroll
                for (int t = 0; t < NTR; ++t) cz[t] = load_win(a.c[t], o, s2);
                z_fluxes<NTR>(g, i, j, k, load_win(pw, o, sx), load_win(pw, o, s1), load_win(pw, o, s2), load_win(pu, o, s2),
                              load_win(pv, o, s2), cz, fz);
            }
            if (close_cell) {
                const int pb = buf ^ 1;
                const long pkm = pk - 1;
                const double vc = g.vinv_c[pkm], vf = g.vinv_f[pkm];
                const unsigned q = o - s2;
                const int km = k - 1;
                double Gn_[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const double dx = FX[pb][f][wave][lane + 1] - FX[pb][f][wave][lane];
                    const double dy = FY[pb][f][wave + 1][lane] - FY[pb][f][wave][lane];
                    const double div = (f == 2 ? vf : vc) * ((dx + dy) + (fz[f] - fz_prev[f]));
                    Gn_[f] = -div + 0.0;
                }
                const bool mu = in_range(a.ru, i, j, km), mv = in_range(a.rv, i, j, km), mw = in_range(a.rw, i, j, km),
                           mc = in_range(a.rc, i, j, km);
                if (mu) st8(a.Gu, q, Gn_[0]);
                if (mv) st8(a.Gv, q, Gn_[1]);
                if (mw) st8(a.Gw, q, Gn_[2]);
#pragma unroll
                for (int t = 0; t < NTR; ++t)
                    if (mc) st8(a.Gc[t], q, Gn_[3 + t]);
                if (a.substep) {
                    // rk3_substep_field! of the next stage on the cell just closed: same operation order as rk3_substep_kernel
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const bool msk = f == 0 ? mu : (f == 1 ? mv : (f == 2 ? mw : mc));
                        if (!msk) continue;
                        const double *Uf = f == 0 ? a.u : (f == 1 ? a.v : (f == 2 ? a.w : a.c[f >= 3 ? f - 3 : 0]));
                        double Uv = ZWIN ? (f == 0 ? uz.s[2] : (f == 1 ? vz.s[2] : (f == 2 ? wz.s[2] : czw[f >= 3 ? f - 3 : 0].s[2])))
                                         : ld8(Uf, q);
                        if (a.has_zeta) Uv += a.dt * (a.gamma * Gn_[f] + a.zeta * gm[f]);
                        else            Uv += a.dt * a.gamma * Gn_[f];
                        st8(a.Un[f], q, Uv);
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) fz_prev[f] = fz[f];
            if (!last) {
                // ---- low-side x- and y-fluxes of plane k -> LDS ----
                double fl[NF];
                {
                    Win6 cx[NA];
#pragma unroll
                    for (int t = 0; t < NTR; ++t) cx[t] = load_win(a.c[t], o, sx);
                    x_fluxes<NTR>(g, i, j, k, axk, g.ax + pk - 2, load_win(pu, o, sx), load_win(pu, o, s1), ZWIN ? uz : load_win(pu, o, s2),
                                  load_win(pv, o, sx), load_win(pw, o, sx), cx, fl);
#pragma unroll
                    for (int f = 0; f < NF; ++f) FX[buf][f][wave][lane] = fl[f];
                }
                {
                    Win6 cy[NA];
#pragma unroll
                    for (int t = 0; t < NTR; ++t) cy[t] = load_win(a.c[t], o, s1);
                    y_fluxes<NTR>(g, i, j, k, ayk, g.ay + pk - 2, load_win(pv, o, sx), load_win(pv, o, s1), ZWIN ? vz : load_win(pv, o, s2),
                                  load_win(pu, o, s1), load_win(pw, o, s1), cy, fl);
#pragma unroll
                    for (int f = 0; f < NF; ++f) FY[buf][f][wave][lane] = fl[f];
                }
            }
        } else if (edge && !last) {
            double fl[NF];
            if (edge_y) {
                // y-fluxes of row j0+TY (this wave's `j`), all 64 columns
                const unsigned o = col + s2 * (unsigned)pk;
                Win6 cy[NA];
#pragma unroll
                for (int t = 0; t < NTR; ++t) cy[t] = load_win(a.c[t], o, s1);
                y_fluxes<NTR>(g, i, j, k, ayk, g.ay + pk - 2, load_win(a.v, o, sx), load_win(a.v, o, s1), load_win(a.v, o, s2),
                              load_win(a.u, o, s1), load_win(a.w, o, s1), cy, fl);
#pragma unroll
                for (int f = 0; f < NF; ++f) FY[buf][f][TY][lane] = fl[f];
            }
            if (edge_x) {
                // x-fluxes of column i0+64, rows j0 .. j0+TY-1 (one lane per row)
                const unsigned o = cole + s2 * (unsigned)pk;
                Win6 cx[NA];
#pragma unroll
                for (int t = 0; t < NTR; ++t) cx[t] = load_win(a.c[t], o, sx);
                x_fluxes<NTR>(g, ie, je, k, axk, g.ax + pk - 2, load_win(a.u, o, sx), load_win(a.u, o, s1), load_win(a.u, o, s2),
                              load_win(a.v, o, sx), load_win(a.w, o, sx), cx, fl);
#pragma unroll
                for (int f = 0; f < NF; ++f) FX[buf][f][lane][64] = fl[f];
            }
        }
        if (!last) __syncthreads();
    }
}

// grids the flux-sharing kernels (this file and ocn_tendency_roles.h) handle
static inline bool fused_tendency_supported(const DGrid &g, const int *range) {
    (void)range;
    return g.tx != 1 && g.tx != 4 && g.tx != 5 && g.ty != 1 && g.ty != 4 && g.ty != 5 &&     // x, y Periodic or FullyConnected (no wall): identical x / y parent extents for all fields
           g.tx != 3 && g.ty != 3 && g.tz != 3 &&   // Flat directions take the per-field kernels
           g.Bx == 3 && g.By == 3 && g.Bz == 3;     // so do directions with an adapted (reduced-order) scheme
}
// the all-fields kernel of this file addresses a parent array with 32-bit byte offsets
static inline bool fused_tendency_size_supported(const DGrid &g) {
    return 8.0 * (g.Nx + 2.0 * g.Hx) * (g.Ny + 2.0 * g.Hy) * (g.Nz + 2.0 * g.Hz + 1.0) < 4294967296.0;
}

// tuned on MI355X at 256^3 (tools/tune_fused.py): 64 x 7 tiles, register z-windows, 2 waves/SIMD (no spills);
// kchunk = 0: levels per workgroup chosen per launch so that the grid fills whole rounds of the chip (see pick_kchunk)
static int g_fused_ty = 7, g_fused_kchunk = 0, g_fused_minw = 2, g_fused_zwin = 1;
static int g_fused_xcd = 0;     // XCD-aware tile order (FusedArgs::xcd_swizzle): measured 1.445 vs 1.440 ms at 256^3 -- no effect, off
static int g_num_cus = 256;

// Each workgroup primes 3 planes before its first cell closes, and the grid runs in rounds of one workgroup per CU (2 waves
// per SIMD): cost ~ rounds x (kchunk + 3). Pick the cheapest kchunk in [12, 64].
static inline int pick_kchunk(int tiles_xy, int nz) {
    int best = 32;
    long best_cost = -1;
    for (int kc = 12; kc <= 64; ++kc) {
        const long blocks = (long)tiles_xy * ((nz + kc - 1) / kc);
        const long rounds = (blocks + g_num_cus - 1) / g_num_cus;
        const long cost = rounds * (std::min(kc, nz) + 3);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = kc; }
    }
    return best;
}


template <int NTR, int TY>
static int launch_fused_t(const DGrid &g, hipStream_t stream, FusedArgs &a) {
    const int nx = a.r.i1 - a.r.i0 + 1, ny = a.r.j1 - a.r.j0 + 1, nz = a.r.k1 - a.r.k0 + 1;
    if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
    if (a.kchunk <= 0) a.kchunk = pick_kchunk(((nx + 63) / 64) * ((ny + TY - 1) / TY), nz);
    dim3 grid((nx + 63) / 64, (ny + TY - 1) / TY, (nz + a.kchunk - 1) / a.kchunk);
    const dim3 blk(64 * (TY + 1));
#define OCN_LAUNCH_FUSED(BZV, MWV, ZWV) hipLaunchKernelGGL((fused_tendency_kernel<NTR, TY, BZV, MWV, ZWV>), grid, blk, 0, stream, g, a)
    // waves per SIMD the register allocator must allow without the register z-windows: two (TY+1)-wave workgroups per CU
    constexpr int MW2 = (2 * (TY + 1) + 3) / 4;
    if (g_fused_zwin) {
        if (g.tz != 0) OCN_LAUNCH_FUSED(true, 2, true); else OCN_LAUNCH_FUSED(false, 2, true);
    } else {
        if (g.tz != 0) OCN_LAUNCH_FUSED(true, MW2, false); else OCN_LAUNCH_FUSED(false, MW2, false);
    }
#undef OCN_LAUNCH_FUSED
    return 0;
}

// optional fused substep: next-stage prognostic arrays, previous tendencies and RK3 coefficients
struct FusedSubstep {
    double *const *Un;           // u, v, w, tracers (3 + ntr)
    const double *const *Gm;
    double dt, gamma, zeta;
    int has_zeta;
    // false on the evaluation that follows RK3's SECOND stage: G(U²) is consumed by the third stage's substep riding along and by nothing
    // else -- no cache_previous_tendencies! follows the third stage (runge_kutta_3.jl:150-166) and the closing update_state! overwrites Gⁿ
    // -- so the kernels that carry that substep do not store it (after a time-step Gⁿ = G(U³), G⁻ = G(U¹), as in the reference)
    bool store_G = true;
};

static inline int launch_fused_tendency(const DGrid &g, hipStream_t stream, const double *u, const double *v, const double *w,
                                        const double *const *tr, int ntr, double *Gu, double *Gv, double *Gw,
                                        double *const *Gc, const int *range, const FusedSubstep *sub = nullptr) {
    FusedArgs a;
    a.u = u; a.v = v; a.w = w; a.Gu = Gu; a.Gv = Gv; a.Gw = Gw;
    for (int t = 0; t < ntr; ++t) { a.c[t] = tr[t]; a.Gc[t] = Gc[t]; }
    a.substep = sub ? 1 : 0;
    a.has_zeta = sub ? sub->has_zeta : 0;
    a.dt = sub ? sub->dt : 0.0; a.gamma = sub ? sub->gamma : 0.0; a.zeta = sub ? sub->zeta : 0.0;
    for (int f = 0; f < 3 + ntr; ++f) { a.Un[f] = sub ? sub->Un[f] : nullptr; a.Gm[f] = sub ? sub->Gm[f] : nullptr; }
    const int Px = g.Nx + 2 * g.Hx, Py = g.Ny + 2 * g.Hy;
    a.s1 = Px;
    a.s2u = (long)Px * Py;
    a.s2w = (long)Px * Py;
    a.off = (g.Hx - 1) + (long)Px * (g.Hy - 1);
    const int ofs = (g.tz != 0 && g.Nz > 1) ? 1 : 0;      // exclude_periphery: w tendencies start at k = 2 on Bounded z
    if (range) {
        a.r = Range6{range[0], range[1], range[2], range[3], range[4], range[5]};
        a.ru = a.rv = a.rw = a.rc = a.r;                  // KernelParameters launches ignore exclude_periphery
    } else {
        a.r = Range6{1, g.Nx, 1, g.Ny, 1, g.Nz};
        a.ru = a.rv = a.rc = a.r;
        a.rw = Range6{1, g.Nx, 1, g.Ny, 1 + ofs, g.Nz};
    }
    a.kchunk = g_fused_kchunk;
    a.xcd_swizzle = g_fused_xcd;
#define OCN_FUSED_CASE(NTR)                                                          \
    case NTR:                                                                        \
        if (g_fused_ty == 3) return launch_fused_t<NTR, 3>(g, stream, a);            \
        return launch_fused_t<NTR, 7>(g, stream, a);
    switch (ntr) {
        OCN_FUSED_CASE(0)
        OCN_FUSED_CASE(1)
        OCN_FUSED_CASE(2)
        OCN_FUSED_CASE(3)
        default: return -2;
    }
#undef OCN_FUSED_CASE
}
