// The one-pass tendency epilogue (ocn_kernels.h: tendency_epilogue_kernel) as a z-march. Same terms, same operations on the same values --
// every flux below is the expression of ClosureCtx::vf.. / closure_divergence with its operands taken from registers -- but each POINT
// quantity is evaluated once instead of once per consumer:
//
//   the viscous flux tensor is symmetric: vf12 at an ffc point is read by u (its y difference) and by v (its x difference), vf13 by u
//   and w, vf23 by v and w. The per-field kernel evaluates 6 fluxes for each of the three velocity tendencies and 6 per tracer, every
//   one with its own ~8 loads: ~200 loads per cell, issue-bound on the address / L1 path (0.66 ms at 256 x 256 x 128, neither VALU nor
//   HBM busy). Here a wave owns 64 consecutive columns of one row j and marches along z:
//     x: the neighbouring column is the neighbouring LANE (DPP wave_shl / wave_shr: 1), lanes 0 and 63 only supply their columns
//        (62 columns written per wave);
//     z: the fluxes through the top face of level k are those through the bottom face of level k + 1, and the level-(k-1) values the
//        z differences need stay in registers;
//     y: the thread loads the rows j - 1, j, j + 1 of a level ONCE and evaluates the row-(j+1) / row-(j-1) fluxes itself (no LDS, no barrier).
//   ~36 loads per cell and level (24 field / coefficient rows, pHY′, the five tendencies and their previous-stage values).
//
// The Flux conditions are NOT applied here (their run-time indexed description of sides and dependencies, inlined once per field, made the
// compiler move the whole argument block to scratch memory): epilogue_flux_shell_kernel below re-does the boundary cells that carry one.
//
// An XCD-contiguous tile order (every XCD a contiguous range of (row block, chunk) tiles, so that the rows shared with the y neighbours sit in one
// L2) was measured: no difference (6.02 / 6.02 ms per step) -- the re-reads are served by the Infinity Cache.
//
// Iteration L loads level L, forms the fluxes that live on level L (own level: vf11, vf22, vf12, the tracers' x / y fluxes, Coriolis and
// pHY′ terms; face level L: vf13, vf23, the tracers' z flux; vf33 of level L - 1) and then completes cell L - 1.
#pragma once

__device__ __forceinline__ double lane_prev(double x) {              // the value lane - 1 holds (lane 0: 0)
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

#define OCN_EPI_MARCH_COLS 62

template <bool COR, bool BUOY, int CLO, int NTR>
__global__ void __launch_bounds__(512) tendency_epilogue_march_kernel(DGrid g, EpilogueArgs a, Range6 R, int kchunk) {
    constexpr bool VAR = CLO == 2;
    constexpr int NT = NTR > 0 ? NTR : 1;
    const int lane = threadIdx.x;                                        // block (64, 4): a wave per row
    const int i = R.i0 + blockIdx.x * OCN_EPI_MARCH_COLS + lane - 1;
    const int j = R.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    if (j > R.j1) return;                                                // rows are independent: no barrier anywhere
    const int kc0 = R.k0 + blockIdx.z * kchunk, kc1 = min(kc0 + kchunk - 1, R.k1);
    const int ic = min(i, R.i1 + 1);                                     // lanes beyond the last needed column repeat it (in bounds)
    const bool out_lane = lane >= 1 && lane <= OCN_EPI_MARCH_COLS && i <= R.i1;
    const FView &u = a.u, &v = a.v, &w = a.w;
    const double rdx = g.rdx, rdy = g.rdy, dx_ = g.dx, dy_ = g.dy, az = dx_ * dy_;
    const bool do_mom = CLO && (VAR || a.nu != 0.0);                     // the per-field kernel's conditions, wave-uniform
    bool do_c[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) do_c[t] = CLO && t < NTR && (VAR || a.kappa[t] != 0.0);

    // ---- level kc0 - 1: what the face level kc0 and vf33(kc0 - 1) need of the level below
    double u_m, v_m, v1_m, w_m, n_m = a.nu, kfcc_m = a.nu, kcfc_m = a.nu, kcfc1_m = a.nu, c_m[NT], k_m[NT];
    {
        const int L = kc0 - 1;
        u_m = u.at(ic, j, L); v_m = v.at(ic, j, L); v1_m = v.at(ic, j + 1, L); w_m = w.at(ic, j, L);
        if (VAR) {
            const double n0 = a.nu_e.at(ic, j - 1, L), n1 = a.nu_e.at(ic, j, L), n2 = a.nu_e.at(ic, j + 1, L);
            n_m = n1; kfcc_m = 0.5 * (lane_prev(n1) + n1); kcfc_m = 0.5 * (n0 + n1); kcfc1_m = 0.5 * (n1 + n2);
        }
#pragma unroll
        for (int t = 0; t < NTR; ++t) { c_m[t] = a.c[t].at(ic, j, L); k_m[t] = VAR ? a.kappa_e[t].at(ic, j, L) : a.kappa[t]; }
    }
    // carried from iteration L - 1: the x + y parts of the divergences of cell L - 1, its Coriolis / pHY′ terms, the face fluxes of level L - 1
    double s_u = 0, s_v = 0, s_w = 0, s_c[NT], cor_u = 0, cor_v = 0, pg_u = 0, pg_v = 0, p13_m = 0, p23_m = 0, p33_m = 0, qz_m[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { s_c[t] = 0; qz_m[t] = 0; }

    for (int L = kc0; L <= kc1 + 1; ++L) {
        const int tz = L - 1 + g.Hz;
        // ---- the rows j - 1, j, j + 1 of level L
        double uL[3], vL[3], wL[3], nL[3], cL[NT][3], kL[NT][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            uL[r] = u.at(ic, j - 1 + r, L); vL[r] = v.at(ic, j - 1 + r, L); wL[r] = w.at(ic, j - 1 + r, L);
            nL[r] = VAR ? a.nu_e.at(ic, j - 1 + r, L) : a.nu;
#pragma unroll
            for (int t = 0; t < NTR; ++t) {
                cL[t][r] = a.c[t].at(ic, j - 1 + r, L);
                kL[t][r] = VAR ? a.kappa_e[t].at(ic, j - 1 + r, L) : a.kappa[t];
            }
        }
        double pL0 = 0, pL1 = 0;
        if (BUOY && L <= kc1) { pL0 = a.pHY.at(ic, j - 1, L); pL1 = a.pHY.at(ic, j, L); }
        // ---- cell L - 1: its tendencies and previous-stage values (needed at the end of this iteration)
        const int k = L - 1;
        const bool fin = L > kc0;
        double Gin[3 + NT], Gmin[3 + NT];
        long qf[3 + NT];
#pragma unroll
        for (int f = 0; f < 3 + NTR; ++f) {
            const FView &fv = f == 0 ? a.u : (f == 1 ? a.v : (f == 2 ? a.w : a.c[f - 3]));
            qf[f] = fv.lin(ic, j, k);
            Gin[f] = fin ? a.Gn[f][qf[f]] : 0.0;
            Gmin[f] = (fin && a.substep && a.has_zeta) ? a.Gm[f][qf[f]] : 0.0;
        }

        // ---- coefficients at the flux locations of level L (abstract_scalar_diffusivity_closure.jl:310-330)
        double kfcc[3], kcfc_j, kcfc_j1;
        if (VAR) {
#pragma unroll
            for (int r = 0; r < 3; ++r) kfcc[r] = 0.5 * (lane_prev(nL[r]) + nL[r]);
            kcfc_j = 0.5 * (nL[0] + nL[1]); kcfc_j1 = 0.5 * (nL[1] + nL[2]);
        } else { kfcc[0] = kfcc[1] = kfcc[2] = kcfc_j = kcfc_j1 = a.nu; }
        const double kfcf = VAR ? 0.5 * (kfcc_m + kfcc[1]) : a.nu;
        const double kcff_j = VAR ? 0.5 * (kcfc_m + kcfc_j) : a.nu, kcff_j1 = VAR ? 0.5 * (kcfc1_m + kcfc_j1) : a.nu;

        // ---- face level L: vf13, vf23 (rows j, j + 1), the tracers' z flux; vf33 of level L - 1
        const double rdzf = g.rdzf[tz], dzf = g.dzf[tz];
        const double w_w = lane_prev(wL[1]);
        const double p13 = -(2 * (kfcf * (0.5 * ((uL[1] - u_m) * rdzf + (wL[1] - w_w) * rdx))));
        const double p23_j = -(2 * (kcff_j * (0.5 * ((vL[1] - v_m) * rdzf + (wL[1] - wL[0]) * rdy))));
        const double p23_j1 = -(2 * (kcff_j1 * (0.5 * ((vL[2] - v1_m) * rdzf + (wL[2] - wL[1]) * rdy))));
        const double p33 = -(2 * (n_m * ((wL[1] - w_m) * g.rdzc[tz - 1])));             // vf33(i, j, L - 1)
        const double sw_new = ((dy_ * dzf) * lane_next(p13) - (dy_ * dzf) * p13) + ((dx_ * dzf) * p23_j1 - (dx_ * dzf) * p23_j);
        double qz[NT];
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            const double kccf = VAR ? 0.5 * (k_m[t] + kL[t][1]) : a.kappa[t];
            qz[t] = -(kccf * ((cL[t][1] - c_m[t]) * rdzf));
        }

        // ---- complete cell k = L - 1
        if (fin) {
            const double vc = g.vinv_c[tz - 1], vf = g.vinv_f[tz - 1];
            double Gout[3 + NT];
            {   // u
                double G = Gin[0];
                if (COR) G = G - cor_u;
                if (BUOY) G = G - pg_u;
                if (do_mom) G = (G - vc * (s_u + (az * p13 - az * p13_m))) + 0.0;
                Gout[0] = G;
            }
            {   // v
                double G = Gin[1];
                if (COR) G = G - cor_v;
                if (BUOY) G = G - pg_v;
                if (do_mom) G = (G - vc * (s_v + (az * p23_j - az * p23_m))) + 0.0;
                Gout[1] = G;
            }
            {   // w: x and y parts from the face level k (carried), vf33(k) from this iteration, vf33(k - 1) carried
                double G = Gin[2];
                if (do_mom) G = (G - vf * (s_w + (az * p33 - az * p33_m))) + 0.0;
                Gout[2] = G;
            }
#pragma unroll
            for (int t = 0; t < NTR; ++t) {
                double G = Gin[3 + t];
                if (do_c[t]) G = (G - vc * (s_c[t] + (az * qz[t] - az * qz_m[t]))) + 0.0;
                Gout[3 + t] = G;
            }
#pragma unroll
            for (int f = 0; f < 3 + NTR; ++f) {
                const Range6 &r = a.r[f];
                if (out_lane && i >= r.i0 && i <= r.i1 && j >= r.j0 && j <= r.j1 && k >= r.k0 && k <= r.k1) {
                    const double G = Gout[f];
                    const int ms = a.store_sides;
                    const bool on_side = ms && (((ms & 1) && i == 1) || ((ms & 2) && i == g.Nx) || ((ms & 4) && j == 1) || ((ms & 8) && j == g.Ny) ||
                                                ((ms & 16) && k == 1) || ((ms & 32) && k == g.Nz));
                    if (a.store_G || on_side) a.Gn[f][qf[f]] = G;
                    if (a.substep) {
                        double Uv = f == 0 ? u_m : (f == 1 ? v_m : (f == 2 ? w_m : c_m[f >= 3 ? f - 3 : 0]));      // the field at (i, j, k): still the "level below"
                        if (a.has_zeta) Uv += a.dt * (a.gamma * G + a.zeta * Gmin[f]);
                        else            Uv += a.dt * a.gamma * G;
                        a.Un[f][qf[f]] = Uv;
                    }
                }
            }
        }

        // ---- own level L (cells of the chunk only): vf11, vf22, vf12, the tracers' x / y fluxes, Coriolis, pHY′
        if (L <= kc1) {
            const double dzc = g.dzc[tz], ax = dy_ * dzc, ay = dx_ * dzc;
            const double u_e = lane_next(uL[1]), v_w = lane_prev(vL[1]), v1_w = lane_prev(vL[2]);
            const double p11 = -(2 * (nL[1] * ((u_e - uL[1]) * rdx)));
            const double p22_j = -(2 * (nL[1] * ((vL[2] - vL[1]) * rdy))), p22_jm = -(2 * (nL[0] * ((vL[1] - vL[0]) * rdy)));
            const double kffc_j = VAR ? 0.5 * (kfcc[0] + kfcc[1]) : a.nu, kffc_j1 = VAR ? 0.5 * (kfcc[1] + kfcc[2]) : a.nu;
            const double p12_j = -(2 * (kffc_j * (0.5 * ((uL[1] - uL[0]) * rdy + (vL[1] - v_w) * rdx))));
            const double p12_j1 = -(2 * (kffc_j1 * (0.5 * ((uL[2] - uL[1]) * rdy + (vL[2] - v1_w) * rdx))));
            s_u = (ax * p11 - ax * lane_prev(p11)) + (ay * p12_j1 - ay * p12_j);
            s_v = (ax * lane_next(p12_j) - ax * p12_j) + (ay * p22_j - ay * p22_jm);
#pragma unroll
            for (int t = 0; t < NTR; ++t) {
                const double kx = VAR ? 0.5 * (lane_prev(kL[t][1]) + kL[t][1]) : a.kappa[t];
                const double ky_j = VAR ? 0.5 * (kL[t][0] + kL[t][1]) : a.kappa[t], ky_j1 = VAR ? 0.5 * (kL[t][1] + kL[t][2]) : a.kappa[t];
                const double qx = -(kx * ((cL[t][1] - lane_prev(cL[t][1])) * rdx));
                const double qy_j = -(ky_j * ((cL[t][1] - cL[t][0]) * rdy)), qy_j1 = -(ky_j1 * ((cL[t][2] - cL[t][1]) * rdy));
                s_c[t] = (ax * lane_next(qx) - ax * qx) + (ay * qy_j1 - ay * qy_j);
            }
            if (COR) {
                const double u_e0 = lane_next(uL[0]);
                cor_u = x_f_cross_U_of(g, a.fcor, [&](int di, int dj) { return dj == 0 ? (di == 0 ? vL[1] : v_w) : (di == 0 ? vL[2] : v1_w); }, i, j, L);
                cor_v = y_f_cross_U_of(g, a.fcor, [&](int di, int dj) { return dj == 0 ? (di == 0 ? uL[1] : u_e) : (di == 0 ? uL[0] : u_e0); }, i, j, L);
            }
            if (BUOY) {
                pg_u = (pL1 - lane_prev(pL1)) * rdx;
                pg_v = (pL1 - pL0) * rdy;
            }
        }
        // ---- level L becomes the level below
        s_w = sw_new; p13_m = p13; p23_m = p23_j; p33_m = p33;
        u_m = uL[1]; v_m = vL[1]; v1_m = vL[2]; w_m = wL[1]; n_m = nL[1];
        kfcc_m = kfcc[1]; kcfc_m = kcfc_j; kcfc1_m = kcfc_j1;
#pragma unroll
        for (int t = 0; t < NTR; ++t) { qz_m[t] = qz[t]; c_m[t] = cL[t][1]; k_m[t] = kL[t][1]; }
    }
}

// The valued / field-dependent Flux conditions after tendency_epilogue_march_kernel: the cells on the sides in `mask` (bit = side, the
// sides that carry a condition for some field) take their condition terms -- epilogue_flux_conditions on the tendency the march left, the
// same call the one-pass kernel makes -- and, when the substep rides along, their next-stage value again from the completed tendency (same
// expression, same operands as the fused form). A cell on several listed sides is handled by the first of them.
struct SideList { int n, side[6]; };
__global__ void __launch_bounds__(256) epilogue_flux_shell_kernel(DGrid g, EpilogueArgs a, int mask, SideList sl) {
    const int s = sl.side[blockIdx.z];                                   // the listed sides only
    const int d = s >> 1;
    const int N[3] = {g.Nx, g.Ny, g.Nz};
    const int ta = blockIdx.x * blockDim.x + threadIdx.x, tb = blockIdx.y * blockDim.y + threadIdx.y;
    if (ta >= (d == 0 ? N[1] : N[0]) || tb >= (d == 2 ? N[1] : N[2])) return;
    const int fixed = (s & 1) ? N[d] : 1;
    const int i = d == 0 ? fixed : ta + 1, j = d == 1 ? fixed : (d == 0 ? ta + 1 : tb + 1), k = d == 2 ? fixed : tb + 1;
    const int idx[3] = {i, j, k};
    for (int e = 0; e < s; ++e)
        if (((mask >> e) & 1) && idx[e >> 1] == ((e & 1) ? N[e >> 1] : 1)) return;
    for (int f = 0; f < a.n; ++f) {
        const Range6 r = a.r[f];
        if (i < r.i0 || i > r.i1 || j < r.j0 || j > r.j1 || k < r.k0 || k > r.k1) continue;
        const FView &fv = f == 0 ? a.u : (f == 1 ? a.v : (f == 2 ? a.w : a.c[f - 3]));
        const long q = fv.lin(i, j, k);
        const double G = epilogue_flux_conditions(g, a, f, i, j, k, q, a.Gn[f][q]);
        a.Gn[f][q] = G;
        if (a.substep) {
            double Uv = fv.p[q];
            if (a.has_zeta) Uv += a.dt * (a.gamma * G + a.zeta * a.Gm[f][q]);
            else            Uv += a.dt * a.gamma * G;
            a.Un[f][q] = Uv;
        }
    }
}
