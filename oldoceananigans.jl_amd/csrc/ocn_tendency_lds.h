// ocn_tendency_lds.h -- the fused flux-sharing tendency kernel (ocn_tendency_fused.h) with the x / y stencil windows served
// from an LDS copy of the current plane's tile instead of ~100 L2 loads per thread and plane.
//
// Why: the register-window kernel is FP64-issue bound but only ~63 % VALU-busy -- at 2 waves/SIMD the L2 latency (500+ cycles)
// of its many small load clusters is not covered. Here every plane of every field is brought on chip ONCE per workgroup:
//   * the (64+6) x (TY+6) halo'd tile of plane k+1 of all fields is prefetched into registers (<= 10 loads per thread) at the top
//     of iteration k, lands in the other LDS tile buffer at the bottom, and is published by the iteration's single s_barrier;
//   * x / y windows are LDS reads (conflict-free: lanes walk consecutive doubles), z windows stay in registers as before;
//   * LDS: 2 tile buffers (72.8 KB) + the double-buffered low-side flux exchange (77.9 KB) = 150.7 KB of the CU's 160 KB,
//     one (TY+1)-wave workgroup per CU = 2 waves/SIMD, the same occupancy the register-window kernel runs at.
// Arithmetic is the shared x_fluxes / y_fluxes / z_fluxes of ocn_tendency_fused.h => bit-identical results.
#pragma once
#include "ocn_tendency_fused.h"

template <int NF, int TY> struct LdsTile {
    static constexpr int W = 70;                 // 64 + 2*3
    static constexpr int R = TY + 6;
    double t[2][NF][R][W];
};

template <int NF, int TY>
__device__ __forceinline__ Win6 lds_win_x(const LdsTile<NF, TY> &T, int b, int f, int ty, int tx) {
    Win6 w;
#pragma unroll
    for (int n = 0; n < 6; ++n) w.s[n] = T.t[b][f][ty][tx - 3 + n];
    return w;
}
template <int NF, int TY>
__device__ __forceinline__ Win6 lds_win_y(const LdsTile<NF, TY> &T, int b, int f, int ty, int tx) {
    Win6 w;
#pragma unroll
    for (int n = 0; n < 6; ++n) w.s[n] = T.t[b][f][ty - 3 + n][tx];
    return w;
}

template <int NTR, int TY, bool BZ>
__global__ void __launch_bounds__(64 * (TY + 1), 2) fused_tendency_lds_kernel(DGrid gin, FusedArgs a) {
    constexpr int NF = 3 + NTR;
    constexpr int NA = NTR > 0 ? NTR : 1;
    constexpr int NT = 64 * (TY + 1);
    constexpr int TW = LdsTile<NF, TY>::W, TR = LdsTile<NF, TY>::R, TE = TW * TR;
    constexpr int NLOAD = (TE + NT - 1) / NT;       // tile elements per thread and field
    __shared__ LdsTile<NF, TY> T;
    __shared__ double FX[2][NF][TY][66];
    __shared__ double FY[2][NF][TY + 1][64];

    DGrid g = gin;
    g.tx = 0; g.ty = 0; g.tz = BZ ? 1 : 0;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i0 = a.r.i0 + blockIdx.x * 64, j0 = a.r.j0 + blockIdx.y * TY;
    const int kc0 = a.r.k0 + blockIdx.z * a.kchunk;
    const int kc1 = min(kc0 + a.kchunk - 1, a.r.k1);
    const bool edge = wave == TY;
    const int i = i0 + lane;
    const int j = j0 + wave;
    const bool cell_ij = !edge && i <= a.r.i1 && j <= a.r.j1;
    const bool flux_ij = !edge && i <= a.r.i1 + 1 && j <= a.r.j1 + 1;
    const bool edge_y = edge && i <= a.r.i1 && j <= a.r.j1 + 1;
    const int ie = i0 + 64, je = j0 + lane;
    const bool edge_x = edge && lane < TY && ie <= a.r.i1 + 1 && je <= a.r.j1;

    const unsigned s1 = 8u * (unsigned)a.s1, s2 = 8u * (unsigned)a.s2u;
    const int Hz = g.Hz;
    const unsigned col = 8u * (unsigned)(a.off + i + (long)a.s1 * j);
    const unsigned cole = 8u * (unsigned)(a.off + ie + (long)a.s1 * je);
    const int tx = lane + 3, ty = wave + 3;           // this thread's cell inside the tile (edge wave: row TY + 3)

    // tile-fill assignment: element e of the TR x TW tile <-> global (i0 - 3 + e % TW, j0 - 3 + e / TW); parent bounds guard
    unsigned tofs[NLOAD];
    int tpos[NLOAD];
    const int imax = g.Nx + g.Hx, jmax = g.Ny + g.Hy;
#pragma unroll
    for (int n = 0; n < NLOAD; ++n) {
        const int e = (int)threadIdx.x + n * NT;
        const int er = e / TW, ec = e - er * TW;
        const int gi = i0 - 3 + ec, gj = j0 - 3 + er;
        const bool ok = e < TE && gi <= imax && gj <= jmax;
        tpos[n] = ok ? e : -1;
        tofs[n] = ok ? 8u * (unsigned)(a.off + gi + (long)a.s1 * gj) : 0u;
    }
    const double *fp[NF];
    fp[0] = a.u; fp[1] = a.v; fp[2] = a.w;
#pragma unroll
    for (int t = 0; t < NTR; ++t) fp[3 + t] = a.c[t];

    // prologue: tile of plane kc0
    {
        const unsigned op = s2 * (unsigned)(kc0 - 1 + Hz);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int n = 0; n < NLOAD; ++n)
                if (tpos[n] >= 0) (&T.t[kc0 & 1][f][0][0])[tpos[n]] = ld8(fp[f], tofs[n] + op);
    }

    double fz_prev[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) fz_prev[f] = 0;

    Win6 uz, vz, wz, czw[NA];
    if (flux_ij) {
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
    __syncthreads();

    for (int k = kc0; k <= kc1 + 1; ++k) {
        const bool last = k == kc1 + 1;
        const int buf = k & 1;
        const long pk = (long)(k - 1 + Hz);
        const double axk = g.ax[pk], ayk = g.ay[pk];
        // ---- prefetch the tile of plane k+1 (needed up to the peeled plane kc1+1, whose z-fluxes read the w windows) ----
        double pre[NF][NLOAD];
        const bool fetch = !last;
        if (fetch) {
            const unsigned op = s2 * (unsigned)(pk + 1);
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int n = 0; n < NLOAD; ++n) pre[f][n] = tpos[n] >= 0 ? ld8(fp[f], tofs[n] + op) : 0.0;
        }
        if (flux_ij) {
            const unsigned o = col + s2 * (unsigned)pk;
            double gm[NF];
            const bool close_cell = k > kc0 && cell_ij;
            if (a.substep && a.has_zeta && close_cell) {
#pragma unroll
                for (int f = 0; f < NF; ++f) gm[f] = ld8(a.Gm[f], o - s2);
            }
            // ---- z-fluxes of plane k, then close cell k-1 ----
            double fz[NF];
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
            z_fluxes<NTR>(g, i, j, k, lds_win_x<NF, TY>(T, buf, 2, ty, tx), lds_win_y<NF, TY>(T, buf, 2, ty, tx), wz, uz, vz, czw, fz);
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
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const bool msk = f == 0 ? mu : (f == 1 ? mv : (f == 2 ? mw : mc));
                        if (!msk) continue;
                        double Uv = f == 0 ? uz.s[2] : (f == 1 ? vz.s[2] : (f == 2 ? wz.s[2] : czw[f >= 3 ? f - 3 : 0].s[2]));
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
                const Win6 ux = lds_win_x<NF, TY>(T, buf, 0, ty, tx), uy = lds_win_y<NF, TY>(T, buf, 0, ty, tx);
                const Win6 vx = lds_win_x<NF, TY>(T, buf, 1, ty, tx), vy = lds_win_y<NF, TY>(T, buf, 1, ty, tx);
                {
                    Win6 cx[NA];
#pragma unroll
                    for (int t = 0; t < NTR; ++t) cx[t] = lds_win_x<NF, TY>(T, buf, 3 + t, ty, tx);
                    x_fluxes<NTR>(g, i, j, k, axk, g.ax + pk - 2, ux, uy, uz, vx, lds_win_x<NF, TY>(T, buf, 2, ty, tx), cx, fl);
#pragma unroll
                    for (int f = 0; f < NF; ++f) FX[buf][f][wave][lane] = fl[f];
                }
                {
                    Win6 cy[NA];
#pragma unroll
                    for (int t = 0; t < NTR; ++t) cy[t] = lds_win_y<NF, TY>(T, buf, 3 + t, ty, tx);
                    y_fluxes<NTR>(g, i, j, k, ayk, g.ay + pk - 2, vx, vy, vz, uy, lds_win_y<NF, TY>(T, buf, 2, ty, tx), cy, fl);
#pragma unroll
                    for (int f = 0; f < NF; ++f) FY[buf][f][wave][lane] = fl[f];
                }
            }
        } else if (edge && !last) {
            double fl[NF];
            if (edge_y) {
                // y-fluxes of row j0+TY (tile row TY+3), all 64 columns; the z-window of v at that row comes from memory
                const unsigned o = col + s2 * (unsigned)pk;
                Win6 cy[NA];
#pragma unroll
                for (int t = 0; t < NTR; ++t) cy[t] = lds_win_y<NF, TY>(T, buf, 3 + t, ty, tx);
                y_fluxes<NTR>(g, i, j, k, ayk, g.ay + pk - 2, lds_win_x<NF, TY>(T, buf, 1, ty, tx), lds_win_y<NF, TY>(T, buf, 1, ty, tx),
                              load_win(a.v, o, s2), lds_win_y<NF, TY>(T, buf, 0, ty, tx), lds_win_y<NF, TY>(T, buf, 2, ty, tx), cy, fl);
#pragma unroll
                for (int f = 0; f < NF; ++f) FY[buf][f][TY][lane] = fl[f];
            }
            if (edge_x) {
                // x-fluxes of column i0+64 (tile column 67), rows j0 .. j0+TY-1 (one lane per row)
                const unsigned o = cole + s2 * (unsigned)pk;
                const int ey = lane + 3, ex = 64 + 3;
                Win6 cx[NA];
#pragma unroll
                for (int t = 0; t < NTR; ++t) cx[t] = lds_win_x<NF, TY>(T, buf, 3 + t, ey, ex);
                x_fluxes<NTR>(g, ie, je, k, axk, g.ax + pk - 2, lds_win_x<NF, TY>(T, buf, 0, ey, ex), lds_win_y<NF, TY>(T, buf, 0, ey, ex),
                              load_win(a.u, o, s2), lds_win_x<NF, TY>(T, buf, 1, ey, ex), lds_win_x<NF, TY>(T, buf, 2, ey, ex), cx, fl);
#pragma unroll
                for (int f = 0; f < NF; ++f) FX[buf][f][lane][64] = fl[f];
            }
        }
        // ---- publish the prefetched tile of plane k+1 (its buffer was last read in iteration k-1, before that barrier) ----
        if (fetch) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int n = 0; n < NLOAD; ++n)
                    if (tpos[n] >= 0) (&T.t[buf ^ 1][f][0][0])[tpos[n]] = pre[f][n];
        }
        if (!last) __syncthreads();
    }
}

template <int NTR, int TY>
static int launch_fused_lds_t(const DGrid &g, hipStream_t stream, FusedArgs &a) {
    const int nx = a.r.i1 - a.r.i0 + 1, ny = a.r.j1 - a.r.j0 + 1, nz = a.r.k1 - a.r.k0 + 1;
    if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
    if (a.kchunk <= 0) a.kchunk = pick_kchunk(((nx + 63) / 64) * ((ny + TY - 1) / TY), nz);
    dim3 grid((nx + 63) / 64, (ny + TY - 1) / TY, (nz + a.kchunk - 1) / a.kchunk);
    const dim3 blk(64 * (TY + 1));
    if (g.tz != 0) hipLaunchKernelGGL((fused_tendency_lds_kernel<NTR, TY, true>), grid, blk, 0, stream, g, a);
    else           hipLaunchKernelGGL((fused_tendency_lds_kernel<NTR, TY, false>), grid, blk, 0, stream, g, a);
    return 0;
}
