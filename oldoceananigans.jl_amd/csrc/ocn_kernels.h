// ocn_kernels.h -- HIP kernels of the hot path (reference kernel inventory: SURVEY.md 2.1). gfx950 / wave64.
#pragma once
#include "ocn_device.h"

// ---------------------------------------------------------------------------------------------------------------------
// advective fluxes (src/Advection/upwind_biased_advective_fluxes.jl:23-121), evaluated straight from global memory.
// This is the "as launched by the reference" formulation (each face flux evaluated by both adjacent cells); the fused
// flux-sharing kernel lives in ocn_tendency_fused.h.
// ---------------------------------------------------------------------------------------------------------------------
enum { AQ_U = 0, AQ_V = 1, AQ_W = 2 };

// area-weighted transports  Ax_qᶠᶜᶜ(u), Ay_qᶜᶠᶜ(v), Az_qᶜᶜᶠ(w)  (Operators/products_between_fields_and_grid_metrics.jl:5-14)
template <int AQ> __device__ __forceinline__ double area_q(const DGrid &g, const FView &f, int i, int j, int k) {
    double a = AQ == AQ_U ? g.ax[k - 1 + g.Hz] : (AQ == AQ_V ? g.ay[k - 1 + g.Hz] : g.az);
    return a * f.at(i, j, k);
}

// symmetric interpolation of a transport along D; CEN: ᶜ variant (stencil of face idx+1). B: buffer of the scheme of the
// direction the FLUX points along (adapt_advection_order; 3 unless that direction has fewer than 3 cells)
template <int AQ, int D, bool CEN>
__device__ __forceinline__ double sym_transport(const DGrid &g, const FView &f, int i, int j, int k, int B) {
    // interpolation along a Flat direction is the identity (flat_advective_fluxes.jl:35-50)
    if ((D == 0 ? g.tx : (D == 1 ? g.ty : g.tz)) == OCN_FLAT) return area_q<AQ>(g, f, i, j, k);
    const int idx = D == 0 ? i : (D == 1 ? j : k);
    const int o = CEN ? 1 : 0;
    double q[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        int m = o - 2 + n;
        if (B < 3 && (n == 0 || n == 3)) { q[n] = 0.0; continue; }      // Centered(order = 2): two points, one halo cell
        q[n] = D == 0 ? area_q<AQ>(g, f, i + m, j, k) : (D == 1 ? area_q<AQ>(g, f, i, j + m, k) : area_q<AQ>(g, f, i, j, k + m));
    }
    if (B < 3) return symmetric_interp_low(q[1], q[2]);
    const int T = D == 0 ? g.tx : (D == 1 ? g.ty : g.tz);
    const bool lo = wall_lo(T), hi = wall_hi(T);
    const int N = D == 0 ? g.Nx : (D == 1 ? g.Ny : g.Nz);
    return symmetric_interp(q[0], q[1], q[2], q[3], lo | hi, idx, CEN, N, lo, hi);
}

template <int D, bool CEN>
__device__ __forceinline__ double biased_field(const DGrid &g, const FView &c, bool left, int i, int j, int k) {
    const int idx = D == 0 ? i : (D == 1 ? j : k);
    const int o = CEN ? 1 : 0;
    const long st = c.stride<D>();
    const double *p = c.p + c.lin(i, j, k) + (o - 3) * st;
    const int T = D == 0 ? g.tx : (D == 1 ? g.ty : g.tz);
    const bool lo = wall_lo(T), hi = wall_hi(T), bounded = lo | hi;
    const int N = D == 0 ? g.Nx : (D == 1 ? g.Ny : g.Nz);
    const int B = D == 0 ? g.Bx : (D == 1 ? g.By : g.Bz);
    if (B < 3) {                                                         // reduced scheme: the halo may be only B cells deep
        const double s2 = p[2 * st], s3 = p[3 * st];
        const double s1 = B == 2 ? p[st] : 0.0, s4 = B == 2 ? p[4 * st] : 0.0;
        return biased_interp_low(s1, s2, s3, s4, left, bounded, idx, CEN, N, B, lo, hi);
    }
    double s0 = p[0], s1 = p[st], s2 = p[2 * st], s3 = p[3 * st], s4 = p[4 * st], s5 = p[5 * st];
    return biased_interp(s0, s1, s2, s3, s4, s5, left, bounded, idx, CEN, N, lo, hi);
}

// advective_momentum_flux_{U,V,W}{u,v,w}: transport AQ interpolated along DS (CS), advected field reconstructed along
// DB (CB).
template <int AQ, int DS, bool CS, int DB, bool CB>
__device__ __forceinline__ double mom_flux(const DGrid &g, const FView &adv, const FView &psi, int i, int j, int k) {
    if ((DB == 0 ? g.tx : (DB == 1 ? g.ty : g.tz)) == OCN_FLAT) return 0.0;     // fluxes along a Flat direction vanish (:13-27)
    double ut = sym_transport<AQ, DS, CS>(g, adv, i, j, k, DB == 0 ? g.Bx : (DB == 1 ? g.By : g.Bz));
    double pr = biased_field<DB, CB>(g, psi, ut > 0, i, j, k);
    return ut * pr;
}

// advective_tracer_flux_{x,y,z} (:99-121): A * U[i,j,k] * cR
template <int D> __device__ __forceinline__ double tracer_flux(const DGrid &g, const FView &vel, const FView &c, int i, int j, int k) {
    if ((D == 0 ? g.tx : (D == 1 ? g.ty : g.tz)) == OCN_FLAT) return 0.0;
    double ut = vel.at(i, j, k);
    double cr = biased_field<D, false>(g, c, ut > 0, i, j, k);
    double a = D == 0 ? g.ax[k - 1 + g.Hz] : (D == 1 ? g.ay[k - 1 + g.Hz] : g.az);
    return a * ut * cr;
}

enum { F_U = 0, F_V = 1, F_W = 2, F_C = 3 };

// compute_Gu!/Gv!/Gw!/Gc! (Models/NonhydrostaticModels/compute_nonhydrostatic_tendencies.jl:138-163) with
// div_𝐯u/v/w (Advection/momentum_advection_operators.jl:46-83) and div_Uc (tracer_advection_operators.jl:29-33).
template <int F>
__global__ void __launch_bounds__(256) tendency_kernel(DGrid g, FView u, FView v, FView w, FView c, FView G, Range6 r) {
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z;
    if (i > r.i1 || j > r.j1 || k > r.k1) return;
    double dx, dy, dz, vinv;
    if (F == F_U) {
        vinv = g.vinv_c[k - 1 + g.Hz];
        dx = mom_flux<AQ_U, 0, true, 0, true>(g, u, u, i, j, k) - mom_flux<AQ_U, 0, true, 0, true>(g, u, u, i - 1, j, k);
        dy = mom_flux<AQ_V, 0, false, 1, false>(g, v, u, i, j + 1, k) - mom_flux<AQ_V, 0, false, 1, false>(g, v, u, i, j, k);
        dz = mom_flux<AQ_W, 0, false, 2, false>(g, w, u, i, j, k + 1) - mom_flux<AQ_W, 0, false, 2, false>(g, w, u, i, j, k);
    } else if (F == F_V) {
        vinv = g.vinv_c[k - 1 + g.Hz];
        dx = mom_flux<AQ_U, 1, false, 0, false>(g, u, v, i + 1, j, k) - mom_flux<AQ_U, 1, false, 0, false>(g, u, v, i, j, k);
        dy = mom_flux<AQ_V, 1, true, 1, true>(g, v, v, i, j, k) - mom_flux<AQ_V, 1, true, 1, true>(g, v, v, i, j - 1, k);
        dz = mom_flux<AQ_W, 1, false, 2, false>(g, w, v, i, j, k + 1) - mom_flux<AQ_W, 1, false, 2, false>(g, w, v, i, j, k);
    } else if (F == F_W) {
        vinv = g.vinv_f[k - 1 + g.Hz];
        dx = mom_flux<AQ_U, 2, false, 0, false>(g, u, w, i + 1, j, k) - mom_flux<AQ_U, 2, false, 0, false>(g, u, w, i, j, k);
        dy = mom_flux<AQ_V, 2, false, 1, false>(g, v, w, i, j + 1, k) - mom_flux<AQ_V, 2, false, 1, false>(g, v, w, i, j, k);
        dz = mom_flux<AQ_W, 2, true, 2, true>(g, w, w, i, j, k) - mom_flux<AQ_W, 2, true, 2, true>(g, w, w, i, j, k - 1);
    } else {
        vinv = g.vinv_c[k - 1 + g.Hz];
        dx = tracer_flux<0>(g, u, c, i + 1, j, k) - tracer_flux<0>(g, u, c, i, j, k);
        dy = tracer_flux<1>(g, v, c, i, j + 1, k) - tracer_flux<1>(g, v, c, i, j, k);
        dz = tracer_flux<2>(g, w, c, i, j, k + 1) - tracer_flux<2>(g, w, c, i, j, k);
    }
    double div = vinv * ((dx + dy) + dz);
    G.at(i, j, k) = -div + 0.0;   // `-div - 0 + 0 ...` of the tendency functions: only maps -0.0 to +0.0
}

// ---------------------------------------------------------------------------------------------------------------------
// ScalarDiffusivity(ν, κ): isotropic, constant, explicit closure (TurbulenceClosures/closure_kernel_operators.jl:
// 22-48, abstract_scalar_diffusivity_closure.jl:194-242, velocity_tracer_gradients.jl:5-42). viscous_flux_* = -2 ν Σᵢⱼ,
// diffusive_flux_* = -κ ∂c, ∂ = δ * (1/Δ); differences along a Flat direction vanish. The kernel ADDS the closure term to a
// tendency that already holds the advective part: G = (G - ∂ⱼτᵢⱼ) + 0.0 -- the order of the terms in
// nonhydrostatic_tendency_kernel_functions.jl:91-100 -- so it composes with the fused advection kernel.
// ---------------------------------------------------------------------------------------------------------------------
struct ClosureCtx {
    const DGrid &g;
    const FView &u, &v, &w;
    double nu;
    bool fx, fy, fz;
    // the coefficient at a flux location: the number `nu`, or -- var -- the ccc array K interpolated there
    // (abstract_scalar_diffusivity_closure.jl:310-330: ν[i,j,k], ℑxyᶠᶠᵃ, ℑxzᶠᵃᶠ, ℑyzᵃᶠᶠ, ℑxᶠᵃᵃ, ℑyᵃᶠᵃ, ℑzᵃᵃᶠ)
    // (held by value: a pointer into the kernel-argument struct would push that struct to scratch memory)
    bool var;
    const FView &K;
    __device__ __forceinline__ double K_ccc(int i, int j, int k) const { return var ? K.at(i, j, k) : nu; }
    __device__ __forceinline__ double K_fcc(int i, int j, int k) const { return var ? 0.5 * (K.at(i - 1, j, k) + K.at(i, j, k)) : nu; }
    __device__ __forceinline__ double K_cfc(int i, int j, int k) const { return var ? 0.5 * (K.at(i, j - 1, k) + K.at(i, j, k)) : nu; }
    __device__ __forceinline__ double K_ccf(int i, int j, int k) const { return var ? 0.5 * (K.at(i, j, k - 1) + K.at(i, j, k)) : nu; }
    __device__ __forceinline__ double K_ffc(int i, int j, int k) const { return var ? 0.5 * (K_fcc(i, j - 1, k) + K_fcc(i, j, k)) : nu; }
    __device__ __forceinline__ double K_fcf(int i, int j, int k) const { return var ? 0.5 * (K_fcc(i, j, k - 1) + K_fcc(i, j, k)) : nu; }
    __device__ __forceinline__ double K_cff(int i, int j, int k) const { return var ? 0.5 * (K_cfc(i, j, k - 1) + K_cfc(i, j, k)) : nu; }
    __device__ __forceinline__ double dzc(int k) const { return g.dzc[k - 1 + g.Hz]; }
    __device__ __forceinline__ double dzf(int k) const { return g.dzf[k - 1 + g.Hz]; }
    // ∂ at Center-in-d (f[+1] - f[0]) and at Face-in-d (f[0] - f[-1]); `δ * Δ⁻¹` (Operators/derivative_operators.jl:20-26) with the reciprocal
    // spacings from the grid's tables: the host formed them as the same IEEE quotients 1.0 / Δ, so a look-up equals the division it replaces
    // (four FP64 divisions per thread of the one-pass epilogue, a third of its arithmetic)
    __device__ __forceinline__ double ddx_c(const FView &f, int i, int j, int k) const { return fx ? 0.0 : (f.at(i + 1, j, k) - f.at(i, j, k)) * g.rdx; }
    __device__ __forceinline__ double ddy_c(const FView &f, int i, int j, int k) const { return fy ? 0.0 : (f.at(i, j + 1, k) - f.at(i, j, k)) * g.rdy; }
    __device__ __forceinline__ double ddz_c(const FView &f, int i, int j, int k) const { return fz ? 0.0 : (f.at(i, j, k + 1) - f.at(i, j, k)) * g.rdzc[k - 1 + g.Hz]; }
    __device__ __forceinline__ double ddx_f(const FView &f, int i, int j, int k) const { return fx ? 0.0 : (f.at(i, j, k) - f.at(i - 1, j, k)) * g.rdx; }
    __device__ __forceinline__ double ddy_f(const FView &f, int i, int j, int k) const { return fy ? 0.0 : (f.at(i, j, k) - f.at(i, j - 1, k)) * g.rdy; }
    __device__ __forceinline__ double ddz_f(const FView &f, int i, int j, int k) const { return fz ? 0.0 : (f.at(i, j, k) - f.at(i, j, k - 1)) * g.rdzf[k - 1 + g.Hz]; }
    __device__ __forceinline__ double S11(int i, int j, int k) const { return ddx_c(u, i, j, k); }
    __device__ __forceinline__ double S22(int i, int j, int k) const { return ddy_c(v, i, j, k); }
    __device__ __forceinline__ double S33(int i, int j, int k) const { return ddz_c(w, i, j, k); }
    __device__ __forceinline__ double S12(int i, int j, int k) const { return 0.5 * (ddy_f(u, i, j, k) + ddx_f(v, i, j, k)); }
    __device__ __forceinline__ double S13(int i, int j, int k) const { return 0.5 * (ddz_f(u, i, j, k) + ddx_f(w, i, j, k)); }
    __device__ __forceinline__ double S23(int i, int j, int k) const { return 0.5 * (ddz_f(v, i, j, k) + ddy_f(w, i, j, k)); }
    // viscous_flux = -2 ν Σ at the location of Σ
    __device__ __forceinline__ double vf11(int i, int j, int k) const { return -(2 * (K_ccc(i, j, k) * S11(i, j, k))); }
    __device__ __forceinline__ double vf22(int i, int j, int k) const { return -(2 * (K_ccc(i, j, k) * S22(i, j, k))); }
    __device__ __forceinline__ double vf33(int i, int j, int k) const { return -(2 * (K_ccc(i, j, k) * S33(i, j, k))); }
    __device__ __forceinline__ double vf12(int i, int j, int k) const { return -(2 * (K_ffc(i, j, k) * S12(i, j, k))); }
    __device__ __forceinline__ double vf13(int i, int j, int k) const { return -(2 * (K_fcf(i, j, k) * S13(i, j, k))); }
    __device__ __forceinline__ double vf23(int i, int j, int k) const { return -(2 * (K_cff(i, j, k) * S23(i, j, k))); }
};

// V⁻¹ (δx(Ax flux) + δy(Ay flux) + δz(Az flux)) of the closure for field F at (i, j, k); coef = ν (momentum) or κ (tracer c)
template <int F>
__device__ __forceinline__ double closure_divergence(const DGrid &g, const FView &u, const FView &v, const FView &w, const FView &c,
                                                     double coef, int i, int j, int k, bool var, const FView &K) {
    // eddy-coefficient arrays come with grids that have no Flat direction (ocn_model_set_amd): with `var` a compile-time constant the
    // three Flat tests fold away and the loads of all six face fluxes are issued together
    const ClosureCtx X{g, u, v, w, coef, var ? false : g.tx == OCN_FLAT, var ? false : g.ty == OCN_FLAT, var ? false : g.tz == OCN_FLAT, var, K};
    const double dx_ = g.dx, dy_ = g.dy;
    double dx, dy, dz, vinv;
    if (F == F_U) {            // ∂ⱼ_τ₁ⱼ at fcc: Ax_qᶜᶜᶜ, Ay_qᶠᶠᶜ, Az_qᶠᶜᶠ
        vinv = g.vinv_c[k - 1 + g.Hz];
        dx = X.fx ? 0.0 : (dy_ * X.dzc(k)) * X.vf11(i, j, k) - (dy_ * X.dzc(k)) * X.vf11(i - 1, j, k);
        dy = X.fy ? 0.0 : (dx_ * X.dzc(k)) * X.vf12(i, j + 1, k) - (dx_ * X.dzc(k)) * X.vf12(i, j, k);
        dz = X.fz ? 0.0 : (dx_ * dy_) * X.vf13(i, j, k + 1) - (dx_ * dy_) * X.vf13(i, j, k);
    } else if (F == F_V) {     // ∂ⱼ_τ₂ⱼ at cfc: Ax_qᶠᶠᶜ, Ay_qᶜᶜᶜ, Az_qᶜᶠᶠ
        vinv = g.vinv_c[k - 1 + g.Hz];
        dx = X.fx ? 0.0 : (dy_ * X.dzc(k)) * X.vf12(i + 1, j, k) - (dy_ * X.dzc(k)) * X.vf12(i, j, k);
        dy = X.fy ? 0.0 : (dx_ * X.dzc(k)) * X.vf22(i, j, k) - (dx_ * X.dzc(k)) * X.vf22(i, j - 1, k);
        dz = X.fz ? 0.0 : (dx_ * dy_) * X.vf23(i, j, k + 1) - (dx_ * dy_) * X.vf23(i, j, k);
    } else if (F == F_W) {     // ∂ⱼ_τ₃ⱼ at ccf: Ax_qᶠᶜᶠ, Ay_qᶜᶠᶠ, Az_qᶜᶜᶜ
        vinv = g.vinv_f[k - 1 + g.Hz];
        dx = X.fx ? 0.0 : (dy_ * X.dzf(k)) * X.vf13(i + 1, j, k) - (dy_ * X.dzf(k)) * X.vf13(i, j, k);
        dy = X.fy ? 0.0 : (dx_ * X.dzf(k)) * X.vf23(i, j + 1, k) - (dx_ * X.dzf(k)) * X.vf23(i, j, k);
        dz = X.fz ? 0.0 : (dx_ * dy_) * X.vf33(i, j, k) - (dx_ * dy_) * X.vf33(i, j, k - 1);
    } else {                   // ∇_dot_qᶜ at ccc: Ax_qᶠᶜᶜ, Ay_qᶜᶠᶜ, Az_qᶜᶜᶠ of -(κ ∂c)
        vinv = g.vinv_c[k - 1 + g.Hz];
        const double ax = dy_ * X.dzc(k), ay = dx_ * X.dzc(k), az = dx_ * dy_;
        dx = X.fx ? 0.0 : ax * -(X.K_fcc(i + 1, j, k) * X.ddx_f(c, i + 1, j, k)) - ax * -(X.K_fcc(i, j, k) * X.ddx_f(c, i, j, k));
        dy = X.fy ? 0.0 : ay * -(X.K_cfc(i, j + 1, k) * X.ddy_f(c, i, j + 1, k)) - ay * -(X.K_cfc(i, j, k) * X.ddy_f(c, i, j, k));
        dz = X.fz ? 0.0 : az * -(X.K_ccf(i, j, k + 1) * X.ddz_f(c, i, j, k + 1)) - az * -(X.K_ccf(i, j, k) * X.ddz_f(c, i, j, k));
    }
    return vinv * ((dx + dy) + dz);
}

template <int F>
__global__ void __launch_bounds__(256) closure_tendency_kernel(DGrid g, FView u, FView v, FView w, FView c, FView G, double coef, Range6 r,
                                                               bool var, FView K) {
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z;
    if (i > r.i1 || j > r.j1 || k > r.k1) return;
    G.at(i, j, k) = (G.at(i, j, k) - closure_divergence<F>(g, u, v, w, c, coef, i, j, k, var, K)) + 0.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// AnisotropicMinimumDissipation(Cν, Cκ; Cb = nothing): eddy viscosity νₑ and eddy diffusivities κₑ at ccc
// (turbulence_closure_implementations/anisotropic_minimum_dissipation.jl:152-196 kernels, :226-333 terms; velocity_tracer_gradients.jl:
// 116-250 normalised gradients -- norm_∂x_u = ∂x_u unscaled, norm_∂x_v = Δᶠx/Δᶠy ∂x_v with Δᶠ = 2Δ at the CALLING index, the
// ℑxzᶜᵃᶜ of norm_∂y_w at :323 kept as written). Sums and products fold left to right, x^2 = x*x, (-C) δ² r / q.
// One thread per cell: νₑ, then κₑ of every tracer (the velocity-gradient terms are shared through the L1/L2).
// ---------------------------------------------------------------------------------------------------------------------
enum AmdOp { A_DXV, A_DYU, A_DXW, A_DZU, A_DYW, A_DZV, A_S12, A_S13, A_S23, A_DXV2, A_DYU2, A_DXW2, A_DZU2, A_DYW2, A_DZV2,
             A_DXV_S12, A_DYU_S12, A_DXW_S13, A_DZU_S13, A_DZV_S23, A_DYW_S23, A_DXC, A_DYC, A_DZC, A_DXC2, A_DYC2, A_DZC2 };
// The derived operands and the interpolations, shared by the two ways the base operands are obtained (CRTP): D::base<OP>(i, j, k)
// returns one of the twelve base operands -- the normalised gradients norm_∂x_v ... and the strain components built from them.
template <class D>
struct AmdTerms {
    __device__ __forceinline__ const D &self() const { return *static_cast<const D *>(this); }
    template <int OP> __device__ __forceinline__ double op(int i, int j, int k) const {
        switch (OP) {
        case A_DXV2: { const double x = op<A_DXV>(i, j, k); return x * x; }
        case A_DYU2: { const double x = op<A_DYU>(i, j, k); return x * x; }
        case A_DXW2: { const double x = op<A_DXW>(i, j, k); return x * x; }
        case A_DZU2: { const double x = op<A_DZU>(i, j, k); return x * x; }
        case A_DYW2: { const double x = op<A_DYW>(i, j, k); return x * x; }
        case A_DZV2: { const double x = op<A_DZV>(i, j, k); return x * x; }
        case A_DXV_S12: return op<A_DXV>(i, j, k) * op<A_S12>(i, j, k);
        case A_DYU_S12: return op<A_DYU>(i, j, k) * op<A_S12>(i, j, k);
        case A_DXW_S13: return op<A_DXW>(i, j, k) * op<A_S13>(i, j, k);
        case A_DZU_S13: return op<A_DZU>(i, j, k) * op<A_S13>(i, j, k);
        case A_DZV_S23: return op<A_DZV>(i, j, k) * op<A_S23>(i, j, k);
        case A_DYW_S23: return op<A_DYW>(i, j, k) * op<A_S23>(i, j, k);
        case A_DXC2: { const double x = op<A_DXC>(i, j, k); return x * x; }
        case A_DYC2: { const double x = op<A_DYC>(i, j, k); return x * x; }
        case A_DZC2: { const double x = op<A_DZC>(i, j, k); return x * x; }
        default: return self().template base<OP>(i, j, k);
        }
    }
    template <int OP> __device__ __forceinline__ double Ix(int i, int j, int k) const { return 0.5 * (op<OP>(i, j, k) + op<OP>(i + 1, j, k)); }
    template <int OP> __device__ __forceinline__ double Iy(int i, int j, int k) const { return 0.5 * (op<OP>(i, j, k) + op<OP>(i, j + 1, k)); }
    template <int OP> __device__ __forceinline__ double Iz(int i, int j, int k) const { return 0.5 * (op<OP>(i, j, k) + op<OP>(i, j, k + 1)); }
    template <int OP> __device__ __forceinline__ double Ixy(int i, int j, int k) const { return 0.5 * (Ix<OP>(i, j, k) + Ix<OP>(i, j + 1, k)); }
    template <int OP> __device__ __forceinline__ double Ixz(int i, int j, int k) const { return 0.5 * (Ix<OP>(i, j, k) + Ix<OP>(i, j, k + 1)); }
    template <int OP> __device__ __forceinline__ double Iyz(int i, int j, int k) const { return 0.5 * (Iy<OP>(i, j, k) + Iy<OP>(i, j, k + 1)); }
};

// metric factors and the direct evaluation of the base operands from the fields
struct AmdFields {
    const DGrid &g;
    const FView &u, &v, &w, &c;
    __device__ __forceinline__ AmdFields(const DGrid &g_, const FView &u_, const FView &v_, const FView &w_, const FView &c_) : g(g_), u(u_), v(v_), w(w_), c(c_) {}
    __device__ __forceinline__ double dzc(int k) const { return g.dzc[k - 1 + g.Hz]; }
    __device__ __forceinline__ double dzf(int k) const { return g.dzf[k - 1 + g.Hz]; }
    __device__ __forceinline__ double FX() const { return 2 * g.dx; }
    __device__ __forceinline__ double FY() const { return 2 * g.dy; }
    __device__ __forceinline__ double FZ(int k) const { return 2 * dzc(k); }
    __device__ __forceinline__ double ddx_c(const FView &f, int i, int j, int k) const { return (f.at(i + 1, j, k) - f.at(i, j, k)) * (1.0 / g.dx); }
    __device__ __forceinline__ double ddy_c(const FView &f, int i, int j, int k) const { return (f.at(i, j + 1, k) - f.at(i, j, k)) * (1.0 / g.dy); }
    __device__ __forceinline__ double ddz_c(const FView &f, int i, int j, int k) const { return (f.at(i, j, k + 1) - f.at(i, j, k)) * (1.0 / dzc(k)); }
    __device__ __forceinline__ double ddx_f(const FView &f, int i, int j, int k) const { return (f.at(i, j, k) - f.at(i - 1, j, k)) * (1.0 / g.dx); }
    __device__ __forceinline__ double ddy_f(const FView &f, int i, int j, int k) const { return (f.at(i, j, k) - f.at(i, j - 1, k)) * (1.0 / g.dy); }
    __device__ __forceinline__ double ddz_f(const FView &f, int i, int j, int k) const { return (f.at(i, j, k) - f.at(i, j, k - 1)) * (1.0 / dzf(k)); }
    template <int OP> __device__ __forceinline__ double direct(int i, int j, int k) const {
        switch (OP) {
        case A_DXV: return FX() / FY() * ddx_f(v, i, j, k);
        case A_DYU: return FY() / FX() * ddy_f(u, i, j, k);
        case A_DXW: return FX() / FZ(k) * ddx_f(w, i, j, k);
        case A_DZU: return FZ(k) / FX() * ddz_f(u, i, j, k);
        case A_DYW: return FY() / FZ(k) * ddy_f(w, i, j, k);
        case A_DZV: return FZ(k) / FY() * ddz_f(v, i, j, k);
        case A_S12: return 0.5 * (direct<A_DYU>(i, j, k) + direct<A_DXV>(i, j, k));
        case A_S13: return 0.5 * (direct<A_DZU>(i, j, k) + direct<A_DXW>(i, j, k));
        case A_S23: return 0.5 * (direct<A_DZV>(i, j, k) + direct<A_DYW>(i, j, k));
        case A_DXC: return FX() * ddx_f(c, i, j, k);
        case A_DYC: return FY() * ddy_f(c, i, j, k);
        default: return FZ(k) * ddz_f(c, i, j, k);     // A_DZC
        }
    }
    __device__ __forceinline__ double delta2(int k) const {
        const double fx = FX(), fy = FY(), fz = FZ(k);
        return 3 / ((1 / (fx * fx) + 1 / (fy * fy)) + 1 / (fz * fz));
    }
};

// every operand recomputed where it is needed (one thread per cell, no cooperation)
struct AmdCtx : AmdFields, AmdTerms<AmdCtx> {
    __device__ __forceinline__ AmdCtx(const DGrid &g_, const FView &u_, const FView &v_, const FView &w_, const FView &c_) : AmdFields(g_, u_, v_, w_, c_) {}
    template <int OP> __device__ __forceinline__ double base(int i, int j, int k) const { return direct<OP>(i, j, k); }
};


template <class A_>
__device__ __forceinline__ double amd_viscosity(const A_ &A, double Cnu, int i, int j, int k) {
    const double dxu = A.ddx_c(A.u, i, j, k), dyv = A.ddy_c(A.v, i, j, k), dzw = A.ddz_c(A.w, i, j, k);
    const double xv2 = A.template Ixy<A_DXV2>(i, j, k), yu2 = A.template Ixy<A_DYU2>(i, j, k), xw2 = A.template Ixz<A_DXW2>(i, j, k), zu2 = A.template Ixz<A_DZU2>(i, j, k),
                 yw2 = A.template Iyz<A_DYW2>(i, j, k), zv2 = A.template Iyz<A_DZV2>(i, j, k);
    double q = dxu * dxu + dyv * dyv;
    q = q + dzw * dzw;
    q = q + xv2; q = q + yu2; q = q + xw2; q = q + zu2; q = q + yw2; q = q + zv2;
    if (q == 0) return fmax(0.0, 0.0);
    double b1 = dxu * (dxu * dxu) + dyv * xv2;
    b1 = b1 + dzw * xw2;
    b1 = b1 + 2 * dxu * A.template Ixy<A_DXV_S12>(i, j, k);
    b1 = b1 + 2 * dxu * A.template Ixz<A_DXW_S13>(i, j, k);
    b1 = b1 + 2 * A.template Ixy<A_DXV>(i, j, k) * A.template Ixz<A_DXW>(i, j, k) * A.template Iyz<A_S23>(i, j, k);
    double b2 = dxu * yu2 + dyv * (dyv * dyv);
    b2 = b2 + dzw * yw2;
    b2 = b2 + 2 * dyv * A.template Ixy<A_DYU_S12>(i, j, k);
    b2 = b2 + 2 * A.template Ixy<A_DYU>(i, j, k) * A.template Iyz<A_DYW>(i, j, k) * A.template Ixz<A_S13>(i, j, k);
    b2 = b2 + 2 * dyv * A.template Iyz<A_DYW_S23>(i, j, k);
    double b3 = dxu * zu2 + dyv * zv2;
    b3 = b3 + dzw * (dzw * dzw);
    b3 = b3 + 2 * A.template Ixz<A_DZU>(i, j, k) * A.template Iyz<A_DZV>(i, j, k) * A.template Ixy<A_S12>(i, j, k);
    b3 = b3 + 2 * dzw * A.template Ixz<A_DZU_S13>(i, j, k);
    b3 = b3 + 2 * dzw * A.template Iyz<A_DZV_S23>(i, j, k);
    const double r = (b1 + b2) + b3;
    const double Cb_zeta = 0.0 / A.FZ(k);
    const double nu = -Cnu * A.delta2(k) * (r - Cb_zeta) / q;
    return fmax(0.0, nu);
}

template <class A_>
__device__ __forceinline__ double amd_diffusivity(const A_ &A, double Ck, int i, int j, int k) {
    const double xc2 = A.template Ix<A_DXC2>(i, j, k), yc2 = A.template Iy<A_DYC2>(i, j, k), zc2 = A.template Iz<A_DZC2>(i, j, k);
    const double sigma = (xc2 + yc2) + zc2;
    if (sigma == 0) return fmax(0.0, 0.0);
    const double dxu = A.ddx_c(A.u, i, j, k), dyv = A.ddy_c(A.v, i, j, k), dzw = A.ddz_c(A.w, i, j, k);
    const double cx = A.template Ix<A_DXC>(i, j, k), cy = A.template Iy<A_DYC>(i, j, k), cz = A.template Iz<A_DZC>(i, j, k);
    double t1 = dxu * xc2 + A.template Ixy<A_DXV>(i, j, k) * cx * cy;
    t1 = t1 + A.template Ixz<A_DXW>(i, j, k) * cx * cz;
    double t2 = A.template Ixy<A_DYU>(i, j, k) * cy * cx + dyv * yc2;
    t2 = t2 + A.template Ixz<A_DYW>(i, j, k) * cy * cz;
    double t3 = A.template Ixz<A_DZU>(i, j, k) * cz * cx + A.template Iyz<A_DZV>(i, j, k) * cz * cy;
    t3 = t3 + dzw * zc2;
    const double theta = (t1 + t2) + t3;
    const double kap = -Ck * A.delta2(k) * theta / sigma;
    return fmax(0.0, kap);
}

struct AmdArgs {
    int ntr;
    Range6 r;
    FView u, v, w, c[OCN_MAX_FIELDS], nu_e, kappa_e[OCN_MAX_FIELDS];
    double Cnu, Ck[OCN_MAX_FIELDS];
};

// Measured at 256 x 256 x 128 (tools/time_amd.py): 0.56 ms at 1 or 2 waves per SIMD (174 VGPRs), 0.51 ms at 3 (<= 168), 0.84 ms at 4
// (spills). Two LDS variants were built, verified bit-identical and dropped: (i) three rotating planes of u, v, w, T, S as LDS tiles
// (~10 global loads per cell instead of ~150): 0.60 ms -- the kernel is not load bound; (ii) the twelve base operands evaluated once
// per patch point and shared through LDS (~550 instead of ~1100 FP64 instructions per cell): 0.65 ms -- the ~350 LDS reads per cell
// then saturate the CU's one LDS pipe (4 clocks per wave-wide 8-byte read) and two barriers per level remain.
#ifndef OCN_AMD_WAVES
#define OCN_AMD_WAVES 3       // waves per SIMD the register allocation must allow
#endif
__global__ void __launch_bounds__(256, OCN_AMD_WAVES) amd_diffusivities_kernel(DGrid g, AmdArgs a) {
    const int i = a.r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = a.r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = a.r.k0 + blockIdx.z;
    if (i > a.r.i1 || j > a.r.j1 || k > a.r.k1) return;
    {
        const AmdCtx A(g, a.u, a.v, a.w, a.u);
        a.nu_e.at(i, j, k) = amd_viscosity(A, a.Cnu, i, j, k);
    }
#pragma unroll 1
    for (int t = 0; t < a.ntr; ++t) {
        const FView c = a.c[t], out = a.kappa_e[t];       // local copies (scalar loads): no address of a kernel argument is taken
        const AmdCtx A(g, a.u, a.v, a.w, c);
        out.at(i, j, k) = amd_diffusivity(A, a.Ck[t], i, j, k);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the same numbers with every POINT operand evaluated once instead of up to four times. Each of the twelve base operands lives
// at a face-face-centre kind of point; a cell's doubly interpolated terms read it at 4 such points (2 for the tracer terms), which it
// shares with its neighbours -- amd_diffusivities_kernel recomputes them all (~1100 FP64 instructions per cell). Here a wave owns a
// 64-wide row of columns and MARCHES along z:
//   x: the operand of the next column is the neighbouring LANE's -- one DPP move per 32 bits (wave_shl:1), no LDS, no recomputation;
//      lane 63 only supplies its column to lane 62 (a wave writes 63 columns);
//   z: the x / y interpolated operands of level k + 1 become those of level k in the next iteration -- registers;
//   y: the thread evaluates the rows j and j + 1 itself (2x instead of 4x; no LDS, no barrier).
// The cell is then assembled by the SAME amd_viscosity / amd_diffusivity templates from a context that returns the stored interpolants,
// so every floating-point operation and its operands are those of the per-cell kernel: bit-identical (tests: == against the oracle and
// against amd_diffusivities_kernel at full size). ~350 FP64 instructions + ~60 lane moves per cell.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_next(double x) {              // the value lane + 1 holds (lane 63: 0)
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ix_lane(double x) { return 0.5 * (x + lane_next(x)); }      // Ix: 0.5 * (op(i) + op(i + 1))

#define OCN_AMD_MAXCHUNK 64
enum { AM_SQ1 = 0, AM_SQ2, AM_P1, AM_P2, AM_A, AM_B, AM_S, AM_N };      // per point kind: a², b², a s, b s, a, b, s = 0.5 (b + a)  [order as listed below]
template <int NTR>
struct AmdMarchCtx : AmdFields {
    // interpolants of the cell being assembled
    double ixy[AM_N];          // ffc kind: DXV2, DYU2, DXV_S12, DYU_S12, DXV, DYU, S12
    double ixz[AM_N + 1];      // fcf kind: DXW2, DZU2, DXW_S13, DZU_S13, DXW, DZU, S13; [7] = Ixz of DYW (anisotropic_minimum_dissipation.jl:323 as written)
    double iyz[AM_N];          // cff kind: DYW2, DZV2, DYW_S23, DZV_S23, DYW, DZV, S23
    double tx, tx2, ty, ty2, tz, tz2;      // tracer being assembled: Ix<DXC>, Ix<DXC2>, Iy<DYC>, Iy<DYC2>, Iz<DZC>, Iz<DZC2>
    // wave-uniform per-level factors of the cell being assembled, evaluated ONCE per block (same expressions as AmdFields::FZ / delta2 --
    // the per-cell kernel repeats these divisions in every thread: 4 per delta2, once per eddy coefficient)
    double fz_k, delta2_k;
    // ∂x u, ∂y v, ∂z w at the cell (AmdFields::ddx_c ... with their reciprocal spacings taken from the table): evaluated once, used by the
    // viscosity and by every tracer's diffusivity
    double dxu_k, dyv_k, dzw_k;
    __device__ __forceinline__ AmdMarchCtx(const DGrid &g_, const FView &u_, const FView &v_, const FView &w_) : AmdFields(g_, u_, v_, w_, u_) {}
    __device__ __forceinline__ double FZ(int) const { return fz_k; }
    __device__ __forceinline__ double delta2(int) const { return delta2_k; }
    __device__ __forceinline__ double ddx_c(const FView &, int, int, int) const { return dxu_k; }
    __device__ __forceinline__ double ddy_c(const FView &, int, int, int) const { return dyv_k; }
    __device__ __forceinline__ double ddz_c(const FView &, int, int, int) const { return dzw_k; }
    template <int OP> __device__ __forceinline__ double Ixy(int, int, int) const {
        return OP == A_DXV2 ? ixy[AM_SQ1] : OP == A_DYU2 ? ixy[AM_SQ2] : OP == A_DXV_S12 ? ixy[AM_P1] : OP == A_DYU_S12 ? ixy[AM_P2]
             : OP == A_DXV ? ixy[AM_A] : OP == A_DYU ? ixy[AM_B] : ixy[AM_S];
    }
    template <int OP> __device__ __forceinline__ double Ixz(int, int, int) const {
        return OP == A_DXW2 ? ixz[AM_SQ1] : OP == A_DZU2 ? ixz[AM_SQ2] : OP == A_DXW_S13 ? ixz[AM_P1] : OP == A_DZU_S13 ? ixz[AM_P2]
             : OP == A_DXW ? ixz[AM_A] : OP == A_DZU ? ixz[AM_B] : OP == A_S13 ? ixz[AM_S] : ixz[AM_N];
    }
    template <int OP> __device__ __forceinline__ double Iyz(int, int, int) const {
        return OP == A_DYW2 ? iyz[AM_SQ1] : OP == A_DZV2 ? iyz[AM_SQ2] : OP == A_DYW_S23 ? iyz[AM_P1] : OP == A_DZV_S23 ? iyz[AM_P2]
             : OP == A_DYW ? iyz[AM_A] : OP == A_DZV ? iyz[AM_B] : iyz[AM_S];
    }
    template <int OP> __device__ __forceinline__ double Ix(int, int, int) const { return OP == A_DXC2 ? tx2 : tx; }
    template <int OP> __device__ __forceinline__ double Iy(int, int, int) const { return OP == A_DYC2 ? ty2 : ty; }
    template <int OP> __device__ __forceinline__ double Iz(int, int, int) const { return OP == A_DZC2 ? tz2 : tz; }
};

// the seven values of a point kind from its two base operands, in AmdTerms::op's forms: a * a, b * b, a * s, b * s, a, b, s = 0.5 * (b + a)
// (S12 = 0.5 (DYU + DXV): a = DXV, b = DYU; S13 = 0.5 (DZU + DXW): a = DXW, b = DZU; S23 = 0.5 (DZV + DYW): a = DYW, b = DZV)
__device__ __forceinline__ void amd_point(double a, double b, double out[AM_N]) {
    const double s = 0.5 * (b + a);
    out[AM_SQ1] = a * a; out[AM_SQ2] = b * b; out[AM_P1] = a * s; out[AM_P2] = b * s; out[AM_A] = a; out[AM_B] = b; out[AM_S] = s;
}

#ifndef OCN_AMD_MARCH_WAVES
#define OCN_AMD_MARCH_WAVES 2       // waves per SIMD the register allocation must allow; 3 (168 VGPRs, 17 doubles spilled): 0.80 ms instead of 0.29 (measured)
#endif
#ifndef OCN_AMD_PF
#define OCN_AMD_PF 0                // levels whose loads are issued ahead of the level being evaluated (measured at 256 x 256 x 128: 2 waves per SIMD PF 0 0.249 ms,
                                    // PF 1 0.64 (256 VGPRs + spills); 1 wave per SIMD PF 1 0.286, PF 2 0.304 -- gathering the loads of a level is what pays, not their distance)
#endif
// every value a level contributes, loaded ONCE per level and ahead of its use: the loads depend on the level only, never on a result, so
// they can be issued OCN_AMD_PF levels early (measured below: no gain over gathering them at the top of their own level)
template <int NTR> struct AmdRaw {
    double uc, vc, wc;                       // u, v, w at (ic, j, L)
    double wim, wjm, wjp, vjp;               // w(ic-1, j), w(ic, j-1), w(ic, j+1), v(ic, j+1)
    double ujp, ujm, uip, vim, vimjp;        // u(ic, j+1), u(ic, j-1), u(ic+1, j), v(ic-1, j), v(ic-1, j+1)
    double c[NTR > 0 ? NTR : 1], cim[NTR > 0 ? NTR : 1], cjm[NTR > 0 ? NTR : 1], cjp[NTR > 0 ? NTR : 1];
};

template <int NTR>
__global__ void __launch_bounds__(256, OCN_AMD_MARCH_WAVES) amd_diffusivities_march_kernel(DGrid g, AmdArgs a, int kchunk) {
    // per-level factors of the levels kc0 .. kc1 + 1 of this block: FZ = 2 Δzᶜ, 1 / Δzᶠ, the four metric ratios of the normalised
    // gradients, δ² -- wave-uniform, ~17 FP64 divisions per cell in the per-cell kernel
    __shared__ double lev[8][OCN_AMD_MAXCHUNK + 2];
    const int lane = threadIdx.x;                                        // block (64, 4): a wave per row
    const int i = a.r.i0 + blockIdx.x * 63 + lane;
    const int j = a.r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int kc0 = a.r.k0 + blockIdx.z * kchunk, kc1 = min(kc0 + kchunk - 1, a.r.k1);
    AmdMarchCtx<NTR> A(g, a.u, a.v, a.w);
    const double fx = A.FX(), fy = A.FY(), rdx = 1.0 / g.dx, rdy = 1.0 / g.dy;
    for (int q = threadIdx.x + 64 * threadIdx.y; q <= kc1 + 1 - kc0; q += 256) {
        const int L = kc0 + q;
        const double fz = 2 * A.dzc(L);
        lev[0][q] = fz; lev[1][q] = 1.0 / A.dzf(L);
        lev[2][q] = fx / fz; lev[3][q] = fz / fx; lev[4][q] = fy / fz; lev[5][q] = fz / fy;
        lev[6][q] = 3 / ((1 / (fx * fx) + 1 / (fy * fy)) + 1 / (fz * fz));
        lev[7][q] = 1.0 / A.dzc(L);
    }
    __syncthreads();
    if (j > a.r.j1) return;                                              // (rows are independent from here on: no further barrier)
    const bool out_lane = lane < 63 && i <= a.r.i1;
    const int ic = min(i, a.r.i1 + 1);                                   // lanes beyond the last needed column repeat it (in bounds)
    const double fxfy = fx / fy, fyfx = fy / fx;
    const FView &u = a.u, &v = a.v, &w = a.w;
    typedef AmdRaw<NTR> Raw;

    auto load = [&](int L) {                                             // levels beyond the chunk's last needed one: that one again (unused)
        Raw r;
        L = min(L, kc1 + 1);
        r.uc = u.at(ic, j, L); r.vc = v.at(ic, j, L); r.wc = w.at(ic, j, L);
        r.wim = w.at(ic - 1, j, L); r.wjm = w.at(ic, j - 1, L); r.wjp = w.at(ic, j + 1, L); r.vjp = v.at(ic, j + 1, L);
        r.ujp = u.at(ic, j + 1, L); r.ujm = u.at(ic, j - 1, L); r.uip = u.at(ic + 1, j, L);
        r.vim = v.at(ic - 1, j, L); r.vimjp = v.at(ic - 1, j + 1, L);
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            const FView &c = a.c[t];
            r.c[t] = c.at(ic, j, L); r.cim[t] = c.at(ic - 1, j, L); r.cjm[t] = c.at(ic, j - 1, L); r.cjp[t] = c.at(ic, j + 1, L);
        }
        return r;
    };

    // level state carried from one iteration to the next (see the section comment)
    double ixy[AM_N], ix_fcf[AM_N + 1], iy_cff[AM_N];
    double t_ix[NTR > 0 ? NTR : 1], t_ix2[NTR > 0 ? NTR : 1], t_iy[NTR > 0 ? NTR : 1], t_iy2[NTR > 0 ? NTR : 1], t_z[NTR > 0 ? NTR : 1], t_z2[NTR > 0 ? NTR : 1];
    double n_fcf[AM_N + 1], n_cff[AM_N], n_z[NTR > 0 ? NTR : 1], n_z2[NTR > 0 ? NTR : 1];

    // fcf / cff kinds and the tracers' z operand at level L (what a cell needs of the level ABOVE it); r: level L, rb: level L - 1
    auto vertical = [&](int L, const Raw &r, const Raw &rb, double fcf[AM_N + 1], double cff[AM_N], double tz[], double tz2[]) {
        const int q = L - kc0;
        const double fz = lev[0][q], rdzf = lev[1][q], fxfz = lev[2][q], fzfx = lev[3][q], fyfz = lev[4][q], fzfy = lev[5][q];
        const double uc = r.uc, vc = r.vc, wc = r.wc;
        {   // (ic, j, L): DXW = FX / FZ * ∂x w, DZU = FZ / FX * ∂z u
            double p[AM_N];
            amd_point(fxfz * ((wc - r.wim) * rdx), fzfx * ((uc - rb.uc) * rdzf), p);
#pragma unroll
            for (int q = 0; q < AM_N; ++q) fcf[q] = ix_lane(p[q]);
        }
        double dyw0;
        {   // rows j and j + 1: DYW = FY / FZ * ∂y w, DZV = FZ / FY * ∂z v
            double p0[AM_N], p1[AM_N];
            const double wn = r.wjp, vn = r.vjp;
            dyw0 = fyfz * ((wc - r.wjm) * rdy);
            amd_point(dyw0, fzfy * ((vc - rb.vc) * rdzf), p0);
            amd_point(fyfz * ((wn - wc) * rdy), fzfy * ((vn - rb.vjp) * rdzf), p1);
#pragma unroll
            for (int q = 0; q < AM_N; ++q) cff[q] = 0.5 * (p0[q] + p1[q]);
        }
        fcf[AM_N] = ix_lane(dyw0);                                       // Ix of DYW at (·, j, L)
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            const double dzc = fz * ((r.c[t] - rb.c[t]) * rdzf);
            tz[t] = dzc; tz2[t] = dzc * dzc;
        }
    };
    // ffc kind and the tracers' x / y operands at level L (what a cell needs of its OWN level)
    auto horizontal = [&](const Raw &r) {
        double p0[AM_N], p1[AM_N];
        const double uc = r.uc, un = r.ujp, vc = r.vc, vn = r.vjp;
        amd_point(fxfy * ((vc - r.vim) * rdx), fyfx * ((uc - r.ujm) * rdy), p0);
        amd_point(fxfy * ((vn - r.vimjp) * rdx), fyfx * ((un - uc) * rdy), p1);
#pragma unroll
        for (int q = 0; q < AM_N; ++q) ixy[q] = 0.5 * (ix_lane(p0[q]) + ix_lane(p1[q]));
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            const double cc = r.c[t];
            const double dxc = fx * ((cc - r.cim[t]) * rdx);
            t_ix[t] = ix_lane(dxc); t_ix2[t] = ix_lane(dxc * dxc);
            const double dyc0 = fy * ((cc - r.cjm[t]) * rdy), dyc1 = fy * ((r.cjp[t] - cc) * rdy);
            t_iy[t] = 0.5 * (dyc0 + dyc1); t_iy2[t] = 0.5 * (dyc0 * dyc0 + dyc1 * dyc1);
        }
    };

    Raw r0, r1, r2, r3;
    {
        const Raw rm = load(kc0 - 1);
        r0 = load(kc0); r1 = load(kc0 + 1);
        if (OCN_AMD_PF >= 1) r2 = load(kc0 + 2);
        if (OCN_AMD_PF >= 2) r3 = load(kc0 + 3);
        vertical(kc0, r0, rm, ix_fcf, iy_cff, t_z, t_z2);
        horizontal(r0);
    }
    for (int k = kc0; k <= kc1; ++k) {
        // r0: level k, r1: level k + 1 (r2, r3: the levels after it, in flight)
        Raw rn;
        if (OCN_AMD_PF >= 1) rn = load(k + 2 + OCN_AMD_PF);
        vertical(k + 1, r1, r0, n_fcf, n_cff, n_z, n_z2);
#pragma unroll
        for (int q = 0; q < AM_N; ++q) { A.ixy[q] = ixy[q]; A.ixz[q] = 0.5 * (ix_fcf[q] + n_fcf[q]); A.iyz[q] = 0.5 * (iy_cff[q] + n_cff[q]); }
        A.ixz[AM_N] = 0.5 * (ix_fcf[AM_N] + n_fcf[AM_N]);
        A.fz_k = lev[0][k - kc0]; A.delta2_k = lev[6][k - kc0];
        A.dxu_k = (r0.uip - r0.uc) * rdx;
        A.dyv_k = (r0.vjp - r0.vc) * rdy;
        A.dzw_k = (r1.wc - r0.wc) * lev[7][k - kc0];
        if (out_lane) a.nu_e.at(i, j, k) = amd_viscosity(A, a.Cnu, i, j, k);
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            A.tx = t_ix[t]; A.tx2 = t_ix2[t]; A.ty = t_iy[t]; A.ty2 = t_iy2[t];
            A.tz = 0.5 * (t_z[t] + n_z[t]); A.tz2 = 0.5 * (t_z2[t] + n_z2[t]);
            if (out_lane) a.kappa_e[t].at(i, j, k) = amd_diffusivity(A, a.Ck[t], i, j, k);
        }
#pragma unroll
        for (int q = 0; q <= AM_N; ++q) ix_fcf[q] = n_fcf[q];
#pragma unroll
        for (int q = 0; q < AM_N; ++q) iy_cff[q] = n_cff[q];
#pragma unroll
        for (int t = 0; t < NTR; ++t) { t_z[t] = n_z[t]; t_z2[t] = n_z2[t]; }
        if (k < kc1) horizontal(r1);
        r0 = r1;
        if (OCN_AMD_PF == 0) r1 = load(k + 2);
        else if (OCN_AMD_PF == 1) { r1 = r2; r2 = rn; }
        else { r1 = r2; r2 = r3; r3 = rn; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Buoyancy: BuoyancyTracer | SeawaterBuoyancy(LinearEquationOfState), gravity along -z.
// _update_hydrostatic_pressure! (Models/NonhydrostaticModels/update_hydrostatic_pressure.jl:12-22) over i = 0:Nx+1, j = 0:Ny+1
// (:43-50): pHY′[Nz] = -z_dot_g_b(Nz+1) Δzᶠ(Nz+1); pHY′[k] = pHY′[k+1] - z_dot_g_b(k+1) Δzᶠ(k+1), z_dot_g_bᶜᶜᶠ = 1 * ℑzᵃᵃᶠ(b)
// (BuoyancyFormulations/g_dot_b.jl:4). One thread per column, coalesced in x.
// ---------------------------------------------------------------------------------------------------------------------
struct BuoyancyArgs {
    int kind;                 // 1: tracer b; 2: g (α T - β S) (linear_equation_of_state.jl:71-73)
    const double *bT, *S;
    double grav, alpha, beta;
};
__device__ __forceinline__ double buoyancy_perturbation(const BuoyancyArgs &B, long q) {
    return B.kind == 1 ? B.bT[q] : B.grav * (B.alpha * B.bT[q] - B.beta * B.S[q]);
}

#ifndef OCN_HYDRO_TB
#define OCN_HYDRO_TB 32
#endif
// KIND (BuoyancyArgs::kind) is compile-time and the levels of a batch are clamped instead of guarded, so that the 2 * TB loads of a batch are
// straight-line code: with the run-time `kind` test and the `k >= 1` guard around each load the compiler waited for every level's pair before
// issuing the next (one memory round trip per level: 0.098 ms at 256 x 256 x 128, one wave per SIMD, nothing to hide it behind).
template <int KIND>
__global__ void __launch_bounds__(256) hydrostatic_pressure_kernel(DGrid g, FView c, BuoyancyArgs B, double *pHY, int i0, int i1, int j0, int j1) {
    const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = j0 + blockIdx.y * blockDim.y + threadIdx.y;
    if (i > i1 || j > j1) return;
    const int Nz = g.Nz;
    const long q1 = c.lin(i, j, 1);                                     // level k at q1 + (k - 1) s2
    auto b_of = [&](double t, double s) { return KIND == 1 ? t : B.grav * (B.alpha * t - B.beta * s); };
    double bk1 = b_of(B.bT[q1 + (long)Nz * c.s2], KIND == 1 ? 0.0 : B.S[q1 + (long)Nz * c.s2]);          // b[Nz + 1]
    double p = 0.0;
    // the recurrence is serial in k but its loads are not: fetch TB levels, then sweep them (few columns => latency-bound otherwise:
    // 66 K columns at 256 x 256 are one wave per SIMD)
    constexpr int TB = OCN_HYDRO_TB;
    for (int k0 = Nz; k0 >= 1; k0 -= TB) {
        double tb[TB], sb[TB];
#pragma unroll
        for (int n = 0; n < TB; ++n) {
            const long q = q1 + (long)(max(k0 - n, 1) - 1) * c.s2;      // below the bottom: level 1 again (in bounds, unused)
            tb[n] = B.bT[q];
            sb[n] = KIND == 1 ? 0.0 : B.S[q];
        }
#pragma unroll
        for (int n = 0; n < TB; ++n) {
            const int k = k0 - n;
            if (k >= 1) {
                const double bk = b_of(tb[n], sb[n]);
                const double zb = 1 * (0.5 * (bk + bk1));
                const double dzf = g.dzf[k + g.Hz];            // Δzᶠ(k+1)
                p = k == Nz ? -zb * dzf : p - zb * dzf;
                pHY[q1 + (long)(k - 1) * c.s2] = p;
                bk1 = bk;
            }
        }
    }
}

// G_u -= ∂xᶠᶜᶜ pHY′, G_v -= ∂yᶜᶠᶜ pHY′ (nonhydrostatic_tendency_kernel_functions.jl:14-19,97,159) on tendencies holding the
// advective part; ranges = the tendency ranges of u and v
__device__ __forceinline__ double hydrostatic_gradient_x(const DGrid &g, const FView &p, int i, int j, int k) {
    return g.tx == OCN_FLAT ? 0.0 : (p.at(i, j, k) - p.at(i - 1, j, k)) * g.rdx;
}
__device__ __forceinline__ double hydrostatic_gradient_y(const DGrid &g, const FView &p, int i, int j, int k) {
    return g.ty == OCN_FLAT ? 0.0 : (p.at(i, j, k) - p.at(i, j - 1, k)) * g.rdy;
}

__global__ void __launch_bounds__(256) hydrostatic_gradient_kernel(DGrid g, FView p, FView Gu, FView Gv, Range6 ru, Range6 rv) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    if (i >= ru.i0 && i <= ru.i1 && j >= ru.j0 && j <= ru.j1 && k >= ru.k0 && k <= ru.k1)
        Gu.at(i, j, k) = Gu.at(i, j, k) - hydrostatic_gradient_x(g, p, i, j, k);
    if (i >= rv.i0 && i <= rv.i1 && j >= rv.j0 && j <= rv.j1 && k >= rv.k0 && k <= rv.k1)
        Gv.at(i, j, k) = Gv.at(i, j, k) - hydrostatic_gradient_y(g, p, i, j, k);
}

// ---------------------------------------------------------------------------------------------------------------------
// coriolis = FPlane(f) (Coriolis/f_plane.jl:48-52): x_f_cross_U = -f active_weighted_ℑxyᶠᶜᶜ(v), y_f_cross_U =
// f active_weighted_ℑxyᶜᶠᶜ(u). The active-weighted average (Operators/interpolation_operators.jl:116-130) divides the four-point
// average by the fraction of non-peripheral nodes (Grids/inactive_node.jl:152-156). G_u -= x_f_cross_U, G_v -= y_f_cross_U on
// tendencies holding the advective part.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool inactive_cell(const DGrid &g, int i, int j, int k) {
    return ((wall_lo(g.tx) && i < 1) || (wall_hi(g.tx) && i > g.Nx)) || ((wall_lo(g.ty) && j < 1) || (wall_hi(g.ty) && j > g.Ny)) ||
           (g.tz == OCN_BOUNDED && (k < 1 || k > g.Nz));
}

// x_f_cross_U at (f, c, c) and y_f_cross_U at (c, f, c). The field is read through `q(di, dj)` = value at (i + di, j + dj, k): from memory
// (the stand-alone kernels) or from the registers of the marching epilogue -- one body for both, so the same operations on the same values.
template <class Q>
__device__ __forceinline__ double x_f_cross_U_of(const DGrid &g, double f, Q q, int i, int j, int k) {
    const bool fx = g.tx == OCN_FLAT, fy = g.ty == OCN_FLAT;
    // ℑxyᶠᶜᵃ = ℑyᵃᶜᵃ(ℑxᶠᵃᵃ ·): X(jj) = 0.5 (q[i-1, jj] + q[i, jj]); 0.5 (X(j) + X(j+1)); not_peripheral at (c, f, c)
    auto np = [&](int ii, int jj) { return !(inactive_cell(g, ii, jj, k) || inactive_cell(g, ii, jj - 1, k)) ? 1.0 : 0.0; };
    auto Xq = [&](int dj) { return fx ? q(0, dj) : 0.5 * (q(-1, dj) + q(0, dj)); };
    auto Xn = [&](int jj) { return fx ? np(i, jj) : 0.5 * (np(i - 1, jj) + np(i, jj)); };
    const double qa = fy ? Xq(0) : 0.5 * (Xq(0) + Xq(1));
    const double an = fy ? Xn(j) : 0.5 * (Xn(j) + Xn(j + 1));
    const double aw = an == 0 ? 0.0 : qa / an;
    return -f * aw;
}
template <class Q>
__device__ __forceinline__ double y_f_cross_U_of(const DGrid &g, double f, Q q, int i, int j, int k) {
    const bool fx = g.tx == OCN_FLAT, fy = g.ty == OCN_FLAT;
    // ℑxyᶜᶠᵃ = ℑyᵃᶠᵃ(ℑxᶜᵃᵃ ·): X(jj) = 0.5 (q[i, jj] + q[i+1, jj]); 0.5 (X(j-1) + X(j)); not_peripheral at (f, c, c)
    auto np = [&](int ii, int jj) { return !(inactive_cell(g, ii, jj, k) || inactive_cell(g, ii - 1, jj, k)) ? 1.0 : 0.0; };
    auto Xq = [&](int dj) { return fx ? q(0, dj) : 0.5 * (q(0, dj) + q(1, dj)); };
    auto Xn = [&](int jj) { return fx ? np(i, jj) : 0.5 * (np(i, jj) + np(i + 1, jj)); };
    const double qa = fy ? Xq(0) : 0.5 * (Xq(-1) + Xq(0));
    const double an = fy ? Xn(j) : 0.5 * (Xn(j - 1) + Xn(j));
    const double aw = an == 0 ? 0.0 : qa / an;
    return f * aw;
}
__device__ __forceinline__ double x_f_cross_U(const DGrid &g, double f, const FView &v, int i, int j, int k) {
    return x_f_cross_U_of(g, f, [&](int di, int dj) { return v.at(i + di, j + dj, k); }, i, j, k);
}
__device__ __forceinline__ double y_f_cross_U(const DGrid &g, double f, const FView &u, int i, int j, int k) {
    return y_f_cross_U_of(g, f, [&](int di, int dj) { return u.at(i + di, j + dj, k); }, i, j, k);
}

__global__ void __launch_bounds__(256) fplane_coriolis_kernel(DGrid g, double f, FView u, FView v, FView Gu, FView Gv, Range6 ru, Range6 rv) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    if (i >= ru.i0 && i <= ru.i1 && j >= ru.j0 && j <= ru.j1 && k >= ru.k0 && k <= ru.k1)
        Gu.at(i, j, k) = Gu.at(i, j, k) - x_f_cross_U(g, f, v, i, j, k);
    if (i >= rv.i0 && i <= rv.i1 && j >= rv.j0 && j <= rv.j1 && k >= rv.k0 && k <= rv.k1)
        Gv.at(i, j, k) = Gv.at(i, j, k) - y_f_cross_U(g, f, u, i, j, k);
}

// ---------------------------------------------------------------------------------------------------------------------
// One pass for everything that follows the advective part of the tendencies (nonhydrostatic_tendency_kernel_functions.jl:91-100,
// 153-162,216-229,286-297): G = (((A - f×U) - ∇pHY′) - ∂ⱼτᵢⱼ) + 0 for every prognostic field, and -- optionally -- the RK3 substep
// of the NEXT stage into the second set of prognostic arrays (the viscous stencils of neighbouring cells still read U). The
// terms are the device functions of the stand-alone kernels above, evaluated in the reference's order => identical bits.
// ---------------------------------------------------------------------------------------------------------------------
#define OCN_EPILOGUE_MAX_LIN 12
struct EpilogueArgs {
    int n, ntr;
    FView u, v, w, c[OCN_MAX_FIELDS], pHY;      // .p of the views = the live fields
    double *Gn[OCN_MAX_FIELDS];
    const double *Gm[OCN_MAX_FIELDS];
    double *Un[OCN_MAX_FIELDS];
    Range6 r[OCN_MAX_FIELDS];
    bool has_coriolis, has_buoyancy, substep, has_zeta;
    bool store_G;                               // false: the completed tendency feeds the substep riding along and nothing else (FusedSubstep::store_G)
    int store_sides;                            // ... except on these sides (bit = side): epilogue_flux_shell_kernel re-does their cells from the stored value
    double fcor, nu, kappa[OCN_MAX_FIELDS], dt, gamma, zeta;
    bool amd;                                   // eddy coefficients from arrays (AnisotropicMinimumDissipation)
    FView nu_e, kappa_e[OCN_MAX_FIELDS];
    // valued Flux boundary conditions (compute_flux_bcs.jl:57-163), applied after the interior terms: [field][side]
    bool any_flux;
    bool has_flux[OCN_MAX_FIELDS][6];
    double flux[OCN_MAX_FIELDS][6];
    const double *flux_arr[OCN_MAX_FIELDS][6];  // array-valued Flux conditions (null: the number above)
    int loc[OCN_MAX_FIELDS][3];
    // linear field-dependent Flux conditions flux = a + b φ (linear_flux_bc_kernel), applied after the valued ones in (field, side)
    // order like the stand-alone launches
    int nlin;
    struct Lin { int f, side, dep; double a, b; } lin[OCN_EPILOGUE_MAX_LIN];
};

// the valued and the linear field-dependent Flux conditions of field f at (i, j, k) (parent index q), applied to a tendency G that holds
// the interior terms
__device__ __forceinline__ double epilogue_flux_conditions(const DGrid &g, const EpilogueArgs &a, int f, int i, int j, int k, long q, double G) {
    if (a.any_flux) {
        // compute_x/y/z_bcs!: G[1] += flux A / V, G[N] -= flux A / V (x, then y, then z as the reference launches them)
        const double dz = a.loc[f][2] == OCN_FACE ? g.dzf[k - 1 + g.Hz] : g.dzc[k - 1 + g.Hz];
        const double vol = (g.dx * g.dy) * dz;
        const int N[3] = {g.Nx, g.Ny, g.Nz}, idx[3] = {i, j, k};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double area = d == 0 ? g.dy * dz : (d == 1 ? g.dx * dz : g.dx * g.dy);
            // tangential point of the condition: (j, k) on west / east, (i, k) on south / north, (i, j) on bottom / top
            const long ab = d == 0 ? (long)(j - 1) + (long)g.Ny * (k - 1) : (d == 1 ? (long)(i - 1) + (long)g.Nx * (k - 1) : (long)(i - 1) + (long)g.Nx * (j - 1));
            if (a.has_flux[f][2 * d] && idx[d] == 1) {
                const double *fa = a.flux_arr[f][2 * d];
                G += (fa ? fa[ab] : a.flux[f][2 * d]) * area / vol;
            }
            if (a.has_flux[f][2 * d + 1] && idx[d] == N[d]) {
                const double *fa = a.flux_arr[f][2 * d + 1];
                G -= (fa ? fa[ab] : a.flux[f][2 * d + 1]) * area / vol;
            }
        }
    }
    for (int n = 0; n < a.nlin; ++n) {
        if (a.lin[n].f != f) continue;
        const int sd = a.lin[n].side, d = sd >> 1, dep = a.lin[n].dep;
        const int idxd = d == 0 ? i : (d == 1 ? j : k), Nd = d == 0 ? g.Nx : (d == 1 ? g.Ny : g.Nz);
        if (idxd != ((sd & 1) ? Nd : 1)) continue;
        const double *dp = dep == 0 ? a.u.p : (dep == 1 ? a.v.p : (dep == 2 ? a.w.p : a.c[dep - 3].p));
        const double dz = a.loc[f][2] == OCN_FACE ? g.dzf[k - 1 + g.Hz] : g.dzc[k - 1 + g.Hz];
        const double vol = (g.dx * g.dy) * dz;
        const double area = d == 0 ? g.dy * dz : (d == 1 ? g.dx * dz : g.dx * g.dy);
        const double phi = dp[q];                      // the dependency sits at the location of the field: same parent index
        const double a_ = a.lin[n].a, b_ = a.lin[n].b;
        const double flux = a_ == 0.0 ? b_ * phi : a_ + b_ * phi;
        if (sd & 1) G -= flux * area / vol;
        else        G += flux * area / vol;
    }
    return G;
}

// COR / BUOY / CLO (0 none, 1 constant ν, κ, 2 eddy-coefficient arrays) are compile-time: the terms of one cell then form ONE basic
// block whose ~40 loads the compiler issues together -- with run-time flags every term was its own block behind a branch and its
// loads waited one after the other (0.94 -> see DESIGN.md for the measured time at 256 x 256 x 128).
template <bool COR, bool BUOY, int CLO>
__global__ void __launch_bounds__(256) tendency_epilogue_kernel(DGrid g, EpilogueArgs a) {
    // 0.66 ms at 256 x 256 x 128 with the configs[4] physics: ~200 loads per cell, bound on the address / L1 path (VALU busy < 50 %, 3.5 TB/s). Grids without
    // a Flat direction run tendency_epilogue_march_kernel (ocn_epilogue_march.h, 0.48 ms) instead; this kernel stays for the others and as its reference.
    const int f = blockIdx.z % a.n;
    const Range6 r = a.r[f];
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z / a.n;
    if (i > r.i1 || j > r.j1 || k > r.k1) return;
    const FView &fv = f == 0 ? a.u : (f == 1 ? a.v : (f == 2 ? a.w : a.c[f - 3]));
    const long q = fv.lin(i, j, k);
    double G = a.Gn[f][q];
    constexpr bool VAR = CLO == 2;
    if (f == 0) {
        if (COR) G = G - x_f_cross_U(g, a.fcor, a.v, i, j, k);
        if (BUOY) G = G - hydrostatic_gradient_x(g, a.pHY, i, j, k);
        if (CLO && (VAR || a.nu != 0.0)) G = (G - closure_divergence<F_U>(g, a.u, a.v, a.w, a.u, a.nu, i, j, k, VAR, a.nu_e)) + 0.0;
    } else if (f == 1) {
        if (COR) G = G - y_f_cross_U(g, a.fcor, a.u, i, j, k);
        if (BUOY) G = G - hydrostatic_gradient_y(g, a.pHY, i, j, k);
        if (CLO && (VAR || a.nu != 0.0)) G = (G - closure_divergence<F_V>(g, a.u, a.v, a.w, a.u, a.nu, i, j, k, VAR, a.nu_e)) + 0.0;
    } else if (f == 2) {
        if (CLO && (VAR || a.nu != 0.0)) G = (G - closure_divergence<F_W>(g, a.u, a.v, a.w, a.u, a.nu, i, j, k, VAR, a.nu_e)) + 0.0;
    } else {
        const double kap = a.kappa[f - 3];
        if (CLO && (VAR || kap != 0.0))
            G = (G - closure_divergence<F_C>(g, a.u, a.v, a.w, a.c[f - 3], kap, i, j, k, VAR, a.kappa_e[f - 3])) + 0.0;
    }
    G = epilogue_flux_conditions(g, a, f, i, j, k, q, G);
    if (a.store_G) a.Gn[f][q] = G;
    if (a.substep) {
        double Uv = fv.p[q];
        if (a.has_zeta) Uv += a.dt * (a.gamma * G + a.zeta * a.Gm[f][q]);
        else            Uv += a.dt * a.gamma * G;
        a.Un[f][q] = Uv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// halo fills (src/BoundaryConditions). One launch handles up to OCN_MAX_FIELDS fields of identical parent shape.
// ---------------------------------------------------------------------------------------------------------------------
struct FieldList {
    double *p[OCN_MAX_FIELDS];
    int n;
};

// _fill_periodic_*_halo! (fill_halo_regions_periodic.jl:5-33) along dimension D of parent arrays of shape (P0,P1,P2):
// parent[i] = parent[N+i], parent[N+H+i] = parent[H+i] for i = 1..H, over the whole extent of the two other dims.
template <int D>
__global__ void __launch_bounds__(256) fill_periodic_kernel(FieldList fl, int P0, int P1, int P2, int N, int H) {
    // thread -> (h, a, b): h in [0, 2H) fastest when D == 0 so that accesses stay contiguous in x
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Pa = D == 0 ? P1 : P0;            // first remaining dim
    const int Pb = D == 2 ? P1 : P2;            // second remaining dim
    long total = (long)2 * H * Pa * Pb;
    if (t >= total) return;
    int h, a, b;
    if (D == 0) { h = t % (2 * H); long r = t / (2 * H); a = r % Pa; b = r / Pa; }
    else        { a = t % Pa; long r = t / Pa; if (D == 1) { h = r % (2 * H); b = r / (2 * H); } else { b = r % Pb; h = r / Pb; } }
    int dst = h < H ? h : N + h;                // 0-based parent index: west i-1 (i=1..H) | east N+H+i-1
    int src = h < H ? N + h : h;                //                       N+i-1            | H+i-1  (h-H+H)
    long s0 = 1, s1 = P0, s2 = (long)P0 * P1;
    long od, os;
    if (D == 0) { od = dst * s0 + a * s1 + b * s2; os = src * s0 + a * s1 + b * s2; }
    else if (D == 1) { od = a * s0 + dst * s1 + b * s2; os = a * s0 + src * s1 + b * s2; }
    else { od = a * s0 + b * s1 + dst * s2; os = a * s0 + b * s1 + src * s2; }
    for (int f = 0; f < fl.n; ++f) fl.p[f][od] = fl.p[f][os];
}

// Triply periodic grids: the three directional fills in ONE launch. Filling z, then y, then x over the whole parent extent
// (the reference's order) leaves every halo cell -- edges and corners included -- equal to the interior cell at the wrapped
// index in each direction, so each halo cell can be written directly from that interior cell: same bits, a third of the
// launches, and the x halos are no longer a separate uncoalesced pass. Thread -> one halo cell of the (P0, P1, P2) parent.
__global__ void __launch_bounds__(256) fill_periodic_xyz_kernel(FieldList fl, int P0, int P1, int P2, int N0, int N1, int N2, int H0,
                                                                int H1, int H2) {
    // halo cells are enumerated as three slabs: all (i, j) of the 2*H2 halo planes; then, for interior k, the 2*H1 halo rows;
    // then, for interior (j, k), the 2*H0 halo columns
    const long nz = (long)P0 * P1 * (2 * H2), ny = (long)P0 * (2 * H1) * N2, nx = (long)(2 * H0) * N1 * N2;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int i, j, k;
    if (t < nz) {
        i = t % P0; const long r = t / P0; j = r % P1; const int hz = r / P1;
        k = hz < H2 ? hz : N2 + hz;
    } else if ((t -= nz) < ny) {
        i = t % P0; const long r = t / P0; const int hy = r % (2 * H1);
        j = hy < H1 ? hy : N1 + hy;
        k = H2 + (int)(r / (2 * H1));
    } else if ((t -= ny) < nx) {
        const int hx = t % (2 * H0); const long r = t / (2 * H0);
        i = hx < H0 ? hx : N0 + hx;
        j = H1 + (int)(r % N1);
        k = H2 + (int)(r / N1);
    } else return;
    // wrapped interior source (0-based parent indices; interior is [H, H + N))
    const int si = i < H0 ? i + N0 : (i >= H0 + N0 ? i - N0 : i);
    const int sj = j < H1 ? j + N1 : (j >= H1 + N1 ? j - N1 : j);
    const int sk = k < H2 ? k + N2 : (k >= H2 + N2 ? k - N2 : k);
    const long od = i + (long)P0 * (j + (long)P1 * k), os = si + (long)P0 * (sj + (long)P1 * sk);
    for (int f = 0; f < fl.n; ++f) fl.p[f][od] = fl.p[f][os];
}

// Bounded directions, ONE halo cell (fill_halo_kernels.jl:69-70), launched over the INTERIOR extent (grid N) of the two
// other dims. Center fields: Flux / default -> mirror (fill_halo_regions_flux.jl:9-27); Value / Gradient -> linear
// extrapolation through the boundary face (fill_halo_regions_value_gradient.jl:7-119). Face fields: Open wall value
// (fill_halo_regions_open.jl:2-7; impenetrable = 0), skipped when fill_open_bcs = false.
struct BcSides {
    int kind[OCN_MAX_FIELDS][2];      // [field][lo | hi], OCN_BC_*
    double value[OCN_MAX_FIELDS][2];
    const double *arr[OCN_MAX_FIELDS][2];   // array-valued condition (getbc(::AbstractArray, i, j), boundary_condition.jl:164) or null
    double dlo, dhi;                  // spacing at the boundary faces (Δ between the first interior and the first halo point)
    // the condition at tangential interior point (a, b) (1-based) of a dense (Na, .) array, or the number
    __device__ __forceinline__ double get(int f, int side, long ab) const {
        const double *p = arr[f][side];
        return p ? p[ab] : value[f][side];
    }
};

// ea_lo / ea_hi (z fill of an x-slab rank's DIFFUSIVITY fields only): also fill the first halo column on a connected x side -- the rank
// evaluates the eddy diffusivities at i = 0 / Nx + 1 itself, and a serial Periodic run's x fill copies the z-filled value into those
// cells (the closure's corner interpolations at the rank edge read them); an array-valued condition is read at the nearest interior point
template <int D>
__global__ void __launch_bounds__(256) fill_bounded_kernel(FieldList fl, BcSides bc, FView view, int Na, int Nb, int N, bool face,
                                                           bool fill_open, bool do_lo = true, bool do_hi = true, int ea_lo = 0, int ea_hi = 0) {   // one-sided: Left / RightConnected x
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Nae = Na + ea_lo + ea_hi;
    if (t >= (long)Nae * Nb) return;
    int a = 1 - ea_lo + t % Nae, b = 1 + t / Nae;
    long lo, ilo, hi, ihi;
    if (D == 0) { lo = view.lin(face ? 1 : 0, a, b); ilo = view.lin(1, a, b); hi = view.lin(N + 1, a, b); ihi = view.lin(N, a, b); }
    else if (D == 1) { lo = view.lin(a, face ? 1 : 0, b); ilo = view.lin(a, 1, b); hi = view.lin(a, N + 1, b); ihi = view.lin(a, N, b); }
    else { lo = view.lin(a, b, face ? 1 : 0); ilo = view.lin(a, b, 1); hi = view.lin(a, b, N + 1); ihi = view.lin(a, b, N); }
    const long ab = (long)(min(max(a, 1), Na) - 1) + (long)Na * (b - 1);
    for (int f = 0; f < fl.n; ++f) {
        double *p = fl.p[f];
        if (!face) {
            const double c1 = p[ilo], cN = p[ihi];
            const int kl = bc.kind[f][0], kh = bc.kind[f][1];
            double h0 = c1, h1 = cN;
            if (kl == OCN_BC_VALUE) h0 = c1 + ((c1 - bc.get(f, 0, ab)) / (bc.dlo / 2)) * (-bc.dlo);
            else if (kl == OCN_BC_GRADIENT) h0 = c1 + bc.get(f, 0, ab) * (-bc.dlo);
            if (kh == OCN_BC_VALUE) h1 = cN + ((bc.get(f, 1, ab) - cN) / (bc.dhi / 2)) * bc.dhi;
            else if (kh == OCN_BC_GRADIENT) h1 = cN + bc.get(f, 1, ab) * bc.dhi;
            if (do_lo) p[lo] = h0;
            if (do_hi) p[hi] = h1;
        } else if (fill_open) {
            if (do_lo) p[lo] = bc.kind[f][0] == OCN_BC_OPEN ? bc.get(f, 0, ab) : 0.0;
            if (do_hi) p[hi] = bc.kind[f][1] == OCN_BC_OPEN ? bc.get(f, 1, ab) : 0.0;
        }
    }
}

// (Periodic | FullyConnected, Periodic, Bounded) grids: the bounded z fill and the periodic y and x fills in ONE launch. The reference
// fills z first (one halo cell below and above, over the interior (i, j) only), then y, then x over the whole extent of the other
// dimensions (boundary_condition_ordering.jl:17-46), which leaves
//   * the two z-boundary planes (Center fields: k = 0 and N+1; Face fields: the wall planes k = 1 and N+1) holding, at EVERY (i, j) of
//     the parent, the boundary formula evaluated in the column at the wrapped interior (i, j);
//   * every other plane -- the deeper z halos included, whose interior (i, j) nobody writes -- holding in its x / y halo cells the
//     value of the wrapped interior (i, j) of the same plane.
// Each such cell is written here directly from those sources: same bits, a third of the launches. `zfill`: the z fill runs (Center
// fields always; Face fields when fill_open_bcs). An x-slab rank (x halos owned by the neighbours) passes H0 = 0, N0 = P0 and XC = Hx:
// its XC outermost columns are left to the exchange -- in the z-boundary planes they only take part in the periodic y copy.
// XZ (diffusivity fields of an x-slab rank, see fill_bounded_kernel): the innermost XZ of the XC neighbour columns take the z formula too.
__global__ void __launch_bounds__(256) fill_periodic_xy_bounded_z_kernel(FieldList fl, BcSides bc, int P0, int P1, int P2, int N0, int N1,
                                                                         int N2, int H0, int H1, int H2, bool face, bool zfill, int XC,
                                                                         int NA, int XZ = 0) {
    const int klo = face ? H2 : H2 - 1, khi = N2 + H2;                       // 0-based parent planes of the z fill
    const long nA = zfill ? (long)P0 * P1 * 2 : 0;
    const int nplanes = P2 - (zfill ? 2 : 0);
    const long rowsB = (long)P0 * (2 * H1), colsB = (long)(2 * H0) * N1, perplane = rowsB + colsB;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nA) {
        const int i = t % P0; const long r = t / P0; const int j = r % P1; const int side = r / P1;
        const int si = i < H0 ? i + N0 : (i >= H0 + N0 ? i - N0 : i);
        const int sj = j < H1 ? j + N1 : (j >= H1 + N1 ? j - N1 : j);
        const long col = si + (long)P0 * sj, plane = (long)P0 * P1;
        const long od = i + (long)P0 * j + plane * (side ? khi : klo);
        if (i < XC - XZ || i >= P0 - XC + XZ) {                             // a neighbour's column: periodic y copy only
            if (sj == j) return;
            const long os = col + plane * (side ? khi : klo);
            for (int f = 0; f < fl.n; ++f) fl.p[f][od] = fl.p[f][os];
            return;
        }
        const long ab = (long)min(max(si - H0 - XC, 0), NA - 1) + (long)NA * (sj - H1);      // tangential interior point (i, j) of the source column
        for (int f = 0; f < fl.n; ++f) {
            double *p = fl.p[f];
            double val;
            if (face) val = bc.kind[f][side] == OCN_BC_OPEN ? bc.get(f, side, ab) : 0.0;
            else {
                const double c = p[col + plane * (side ? N2 + H2 - 1 : H2)];         // first / last interior cell of the column
                const int kd = bc.kind[f][side];
                val = c;
                if (side == 0) {
                    if (kd == OCN_BC_VALUE) val = c + ((c - bc.get(f, 0, ab)) / (bc.dlo / 2)) * (-bc.dlo);
                    else if (kd == OCN_BC_GRADIENT) val = c + bc.get(f, 0, ab) * (-bc.dlo);
                } else {
                    if (kd == OCN_BC_VALUE) val = c + ((bc.get(f, 1, ab) - c) / (bc.dhi / 2)) * bc.dhi;
                    else if (kd == OCN_BC_GRADIENT) val = c + bc.get(f, 1, ab) * bc.dhi;
                }
            }
            p[od] = val;
        }
        return;
    }
    t -= nA;
    if (t >= perplane * nplanes) return;
    int kp = (int)(t / perplane);
    if (zfill) { if (kp >= klo) ++kp; if (kp >= khi) ++kp; }
    long q = t % perplane;
    int i, j;
    if (q < rowsB) { i = q % P0; const int hy = q / P0; j = hy < H1 ? hy : N1 + hy; }
    else { q -= rowsB; const int hx = q % (2 * H0); i = hx < H0 ? hx : N0 + hx; j = H1 + (int)(q / (2 * H0)); }
    const int si = i < H0 ? i + N0 : (i >= H0 + N0 ? i - N0 : i);
    const int sj = j < H1 ? j + N1 : (j >= H1 + N1 ? j - N1 : j);
    const long plane = (long)P0 * P1 * kp;
    const long od = i + (long)P0 * j + plane, os = si + (long)P0 * sj + plane;
    for (int f = 0; f < fl.n; ++f) fl.p[f][od] = fl.p[f][os];
}

// compute_x/y/z_bcs! (compute_flux_bcs.jl:57-163): G[1] += flux * A / V, G[N] -= flux * A / V over the interior extent of
// the two other dims. area / volume evaluated per cell by the caller-provided metric look-ups.
template <int D>
__global__ void __launch_bounds__(256) flux_bc_kernel(DGrid g, FView G, int Na, int Nb, int N, int lx, int ly, int lz, bool has_lo,
                                                      double flo, bool has_hi, double fhi, const double *alo, const double *ahi) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Na * Nb) return;
    const int a = 1 + t % Na, b = 1 + t / Na;
    for (int side = 0; side < 2; ++side) {
        if (side == 0 ? !has_lo : !has_hi) continue;
        const int n = side ? N : 1;
        const int i = D == 0 ? n : a, j = D == 1 ? n : (D == 0 ? a : b), k = D == 2 ? n : b;
        const double dx = g.dx, dy = g.dy;          // regular in x, y at every location
        const double dz = lz == OCN_FACE ? g.dzf[k - 1 + g.Hz] : g.dzc[k - 1 + g.Hz];
        const double vol = (dx * dy) * dz;          // volume = Az * Δz
        const double area = D == 0 ? dy * dz : (D == 1 ? dx * dz : dx * dy);
        const double *arr = side ? ahi : alo;
        const double flux = arr ? arr[(long)(a - 1) + (long)Na * (b - 1)] : (side ? fhi : flo);
        double &Gq = G.at(i, j, k);
        if (side) Gq -= flux * area / vol;
        else      Gq += flux * area / vol;
    }
    (void)lx; (void)ly;
}

// One side of a field-dependent Flux condition of the linear family: flux = a + b φ[i, j, k_boundary] (getbc of a
// ContinuousBoundaryFunction with field_dependencies = :φ, continuous_boundary_function.jl:128-161; φ at the location of the field that
// carries the condition => identity interpolation), applied like the valued Flux above.
template <int D>
__global__ void __launch_bounds__(256) linear_flux_bc_kernel(DGrid g, FView G, FView P, int Na, int Nb, int N, int lz, int side, double a_,
                                                             double b_) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)Na * Nb) return;
    const int a = 1 + t % Na, b = 1 + t / Na;
    const int n = side ? N : 1;
    const int i = D == 0 ? n : a, j = D == 1 ? n : (D == 0 ? a : b), k = D == 2 ? n : b;
    const double dx = g.dx, dy = g.dy;
    const double dz = lz == OCN_FACE ? g.dzf[k - 1 + g.Hz] : g.dzc[k - 1 + g.Hz];
    const double vol = (dx * dy) * dz;
    const double area = D == 0 ? dy * dz : (D == 1 ? dx * dz : dx * dy);
    const double phi = P.at(i, j, k);
    const double flux = a_ == 0.0 ? b_ * phi : a_ + b_ * phi;
    double &Gq = G.at(i, j, k);
    if (side) Gq -= flux * area / vol;
    else      Gq += flux * area / vol;
}

// ---------------------------------------------------------------------------------------------------------------------
// RK3 substep / tendency caching (src/TimeSteppers/runge_kutta_3.jl:212-226, store_tendencies.jl:6-9)
// ---------------------------------------------------------------------------------------------------------------------
struct SubstepArgs {
    double *U[OCN_MAX_FIELDS];
    const double *Gn[OCN_MAX_FIELDS];
    const double *Gm[OCN_MAX_FIELDS];
    FView view[OCN_MAX_FIELDS];     // .p unused
    Range6 r[OCN_MAX_FIELDS];
    int n;
};

__global__ void __launch_bounds__(256) rk3_substep_kernel(SubstepArgs a, double dt, double gamma, double zeta, bool has_zeta) {
    const int f = blockIdx.z % a.n;
    const Range6 r = a.r[f];
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z / a.n;
    if (i > r.i1 || j > r.j1 || k > r.k1) return;
    const long q = a.view[f].lin(i, j, k);
    double Uv = a.U[f][q];
    if (has_zeta) Uv += dt * (gamma * a.Gn[f][q] + zeta * a.Gm[f][q]);
    else          Uv += dt * gamma * a.Gn[f][q];
    a.U[f][q] = Uv;
}

// ab2_step_field! (TimeSteppers/quasi_adams_bashforth_2.jl:160-173): Gu = (1.5 + χ) Gⁿ - (0.5 + χ) G⁻ * not_euler; u += Δt Gu
__global__ void __launch_bounds__(256) ab2_step_kernel(SubstepArgs a, double dt, double chi) {
    const int f = blockIdx.z % a.n;
    const Range6 r = a.r[f];
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z / a.n;
    if (i > r.i1 || j > r.j1 || k > r.k1) return;
    const long q = a.view[f].lin(i, j, k);
    const bool not_euler = chi != -0.5;            // `* false` is a strong zero: leftover NaNs in G⁻ cannot leak into a Euler step
    const double prev = not_euler ? (0.5 + chi) * a.Gm[f][q] * 1.0 : 0.0;
    const double Gu = (1.5 + chi) * a.Gn[f][q] - prev;
    a.U[f][q] += dt * Gu;
}

__global__ void __launch_bounds__(256) cache_tendencies_kernel(SubstepArgs a) {
    const int f = blockIdx.z % a.n;
    const Range6 r = a.r[f];
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z / a.n;
    if (i > r.i1 || j > r.j1 || k > r.k1) return;
    const long q = a.view[f].lin(i, j, k);
    a.U[f][q] = a.Gn[f][q];
}

// ---------------------------------------------------------------------------------------------------------------------
// pressure: source term, correction, scaling (src/Models/NonhydrostaticModels/{solve_for_pressure,pressure_correction}.jl)
// ---------------------------------------------------------------------------------------------------------------------
// the source term of the pressure equation at cell (i, j, k): divᶜᶜᶜ(u*) [x Δzᶜ for the Fourier-tridiagonal solver]
// (solve_for_pressure.jl:12-84, Operators/divergence_operators.jl:16-19)
// wrap: bit 0 / 1 / 2 = the upper x / y / z neighbour of the last cell is read at its wrapped interior index instead of the halo;
// ue (x-slab ranks): u[Nx + 1, j, k] is read from this dense (Ny, Nz) buffer -- the east neighbour's first column, just received
__device__ __forceinline__ double source_value(const DGrid &g, const FView &u, const FView &v, const FView &w, int i, int j, int k,
                                               bool weight_by_dz, int wrap, const double *ue = nullptr) {
    const int kk = k - 1 + g.Hz;
    const double ax = g.ax[kk], ay = g.ay[kk], az = g.az;
    const int ip = ((wrap & 1) && i == g.Nx) ? 1 : i + 1, jp = ((wrap & 2) && j == g.Ny) ? 1 : j + 1, kp = ((wrap & 4) && k == g.Nz) ? 1 : k + 1;
    const double up = (ue && i == g.Nx) ? ue[(long)(j - 1) + (long)g.Ny * (k - 1)] : u.at(ip, j, k);
    // δ along a Flat direction is zero(FT) (Operators/difference_operators.jl:30-49)
    double dx = g.tx == OCN_FLAT ? 0.0 : ax * up - ax * u.at(i, j, k);      // δxᶜᶜᶜ(Ax_qᶠᶜᶜ, u)
    double dy = g.ty == OCN_FLAT ? 0.0 : ay * v.at(i, jp, k) - ay * v.at(i, j, k);
    double dz = g.tz == OCN_FLAT ? 0.0 : az * w.at(i, j, kp) - az * w.at(i, j, k);
    double div = g.vinv_c[kk] * ((dx + dy) + dz);                 // divᶜᶜᶜ, Operators/divergence_operators.jl:16-19
    return weight_by_dz ? (1.0 * g.dzc[kk]) * div : 1.0 * div;
}

template <bool REAL_OUT>
// rhs element (i, j, k) is stored at (i-1) + sj*(j-1) + sk*(k-1); `pad` (real output only): the row has one extra, zero,
// element after i = Nx (odd local Nx on the distributed solver's paired-column layout)
// wrap (triply periodic grids only): the upper neighbours are read at their wrapped INTERIOR index instead of the halo, so the
// velocity halo fill that precedes the solve in the reference can be left to the next update_state! (same values by construction)
__global__ void __launch_bounds__(256) source_term_kernel(DGrid g, FView u, FView v, FView w, void *rhs, bool weight_by_dz,
                                                          long sj, long sk, bool pad, int wrap = 0, const double *ue = nullptr) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    const double val = source_value(g, u, v, w, i, j, k, weight_by_dz, wrap, ue);
    const long q = (long)(i - 1) + sj * (j - 1) + sk * (k - 1);
    if (REAL_OUT) {                                         // real-transform paths (rhs is real by construction)
        ((double *)rhs)[q] = val;
        if (pad && i == g.Nx) ((double *)rhs)[q + 1] = 0.0;
    } else ((double2 *)rhs)[q] = make_double2(val, 0.0);    // the reference's complex storage
}

// pdiv != nullptr: also `pNHS ./= Δt⁺`, written to a SECOND haloed array (neighbouring threads still read p) that the caller swaps in
__global__ void __launch_bounds__(256) pressure_correction_kernel(DGrid g, FView u, FView v, FView w, FView p, Range6 r,
                                                                  double *pdiv = nullptr, double divisor = 1.0) {
    const int i = r.i0 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = r.j0 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = r.k0 + blockIdx.z;
    if (i > r.i1 || j > r.j1 || k > r.k1) return;
    const double pc = p.at(i, j, k);
    if (pdiv) pdiv[p.lin(i, j, k)] = pc / divisor;
    u.at(i, j, k) -= (g.tx == OCN_FLAT ? 0.0 : pc - p.at(i - 1, j, k)) * g.rdx;            // ∂xᶠᶜᶜ = δx * Δx⁻¹
    v.at(i, j, k) -= (g.ty == OCN_FLAT ? 0.0 : pc - p.at(i, j - 1, k)) * g.rdy;
    w.at(i, j, k) -= (g.tz == OCN_FLAT ? 0.0 : pc - p.at(i, j, k - 1)) * g.rdzf[k - 1 + g.Hz];
}

// _make_pressure_correction! and `pNHS ./= Δt⁺` in one pass, with the pressure read from the DENSE real array the inverse transform
// leaves (x, y Periodic: the lower neighbours wrap; z Periodic wraps, z Bounded has p[0] = p[1] -- the no-flux halo -- so the
// bottom face keeps its w). The haloed pressure field receives p / Δt⁺. Same expressions as pressure_correction_kernel +
// divide_interior_kernel.
__global__ void __launch_bounds__(256) pressure_correction_dense_kernel(DGrid g, FView u, FView v, FView w, const double *pd, FView p,
                                                                        double divisor, bool zbounded, bool store_p) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    const long sx = g.Nx, sxy = (long)g.Nx * g.Ny;
    const long q = (long)(i - 1) + sx * (j - 1) + sxy * (k - 1);
    const double pc = pd[q];
    const double pim = pd[i == 1 ? q + (g.Nx - 1) : q - 1];
    const double pjm = pd[j == 1 ? q + sx * (g.Ny - 1) : q - sx];
    const double pkm = k == 1 ? (zbounded ? pc : pd[q + sxy * (g.Nz - 1)]) : pd[q - sxy];
    u.at(i, j, k) -= (pc - pim) * g.rdx;
    v.at(i, j, k) -= (pc - pjm) * g.rdy;
    w.at(i, j, k) -= (pc - pkm) * g.rdzf[k - 1 + g.Hz];
    if (store_p) p.at(i, j, k) = pc / divisor;
}

__global__ void __launch_bounds__(256) divide_interior_kernel(DGrid g, FView p, double divisor) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    p.at(i, j, k) /= divisor;
}

// ---------------------------------------------------------------------------------------------------------------------
// Poisson solvers (src/Solvers)
// ---------------------------------------------------------------------------------------------------------------------
// `@. ϕc = -b / (λx + λy + λz - m)` with m = 0 and `ϕc[1,1,1] = 0` (fft_based_poisson_solver.jl:110,115)
// Nxs = stored row length (Nx for complex-to-complex, Nx/2+1 for the Hermitian half spectrum of the real transform);
// `scale` folds the inverse-FFT normalisation 1/prod(N) into the same pass when apply_scale is set.
__global__ void __launch_bounds__(256) spectral_divide_kernel(double2 *b, const double *lx, const double *ly, const double *lz,
                                                              int Nxs, int Ny, int Nz, double scale, bool apply_scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = blockIdx.z;
    if (i >= Nxs || j >= Ny || k >= Nz) return;
    const long q = (long)i + (long)Nxs * (j + (long)Ny * k);
    double lam = (lx[i] + ly[j]) + lz[k] - 0.0;
    double2 val = b[q];
    val.x = -val.x / lam;
    val.y = -val.y / lam;
    if (apply_scale) { val.x *= scale; val.y *= scale; }
    if (q == 0) { val.x = 0.0; val.y = 0.0; }
    b[q] = val;
}

// copy_real_component! (fft_based_poisson_solver.jl:129-137) fused with the ifft normalisation (AbstractFFTs ScaledPlan:
// multiply by 1/prod(N)) and, for the Fourier-tridiagonal solver, the mean removal
// `ϕ .= ϕ .- mean(ϕ)` (fourier_tridiagonal_poisson_solver.jl:233): real(ϕ*scale - mean).
__global__ void __launch_bounds__(256) copy_real_kernel(DGrid g, FView phi, const double2 *src, double scale, bool apply_scale,
                                                        const double2 *mean) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    double val = src[(long)(i - 1) + (long)g.Nx * ((j - 1) + (long)g.Ny * (k - 1))].x;
    if (apply_scale) val *= scale;
    if (mean) val -= mean->x;
    phi.at(i, j, k) = val;
}

// solve_batched_tridiagonal_system_kernel! z direction (batched_tridiagonal_solver.jl:213-245): one thread per (i, j)
// column, coalesced across i. Complex f / phi, real a, b (3-D), c, scratch t (3-D real).
// Nxs columns per row are solved; f / phi have row length Nxs, the real coefficient arrays b / t row length ldb
// (ldb = Nx > Nxs = Nx/2+1 when only the Hermitian half spectrum is solved). fscale multiplies the right-hand side
// (1 for the reference semantics; the folded inverse-FFT normalisation on the real-transform path).
__global__ void __launch_bounds__(64) tridiagonal_z_kernel(int Nxs, int ldb, int Ny, int Nz, const double *__restrict__ a,
                                                           const double *__restrict__ b, const double *__restrict__ c,
                                                           const double2 *__restrict__ f, double *__restrict__ t,
                                                           double2 *__restrict__ phi, double fscale, bool apply_scale, int ldf = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= Nxs || j >= Ny) return;
    const int ldx = ldf > 0 ? ldf : Nxs;              // row pitch of f / phi
    const long st = (long)ldx * Ny, stb = (long)ldb * Ny;
    long q = (long)i + (long)ldx * j, qb = (long)i + (long)ldb * j;
    double beta = b[qb];
    double2 f1 = f[q];
    if (apply_scale) { f1.x *= fscale; f1.y *= fscale; }
    double2 prev = make_double2(f1.x / beta, f1.y / beta);
    phi[q] = prev;
    // The recurrences are serial in k, but every load is independent of them: with < 1 wave per SIMD (Nxs * Ny columns only)
    // the kernel is pure memory latency unless the loads of several levels are in flight together => blocks of TB levels are
    // fetched into registers first, then swept.
    constexpr int TB = 8;
    for (int k0 = 1; k0 < Nz; k0 += TB) {
        double2 fb[TB], ob[TB];
        double bb[TB], cb[TB], ab[TB];
#pragma unroll
        for (int n = 0; n < TB; ++n) {
            const int k = k0 + n;
            if (k < Nz) {
                fb[n] = f[q + (long)(n + 1) * st]; ob[n] = phi[q + (long)(n + 1) * st]; bb[n] = b[qb + (long)(n + 1) * stb];
                cb[n] = c[k - 1]; ab[n] = a[k - 1];
            }
        }
#pragma unroll
        for (int n = 0; n < TB; ++n) {
            const int k = k0 + n;
            if (k < Nz) {
                q += st; qb += stb;
                const double ck1 = cb[n], ak1 = ab[n], bk = bb[n];
                const double tk = ck1 / beta;
                t[qb] = tk;
                beta = bk - ak1 * tk;
                double2 fk = fb[n];
                if (apply_scale) { fk.x *= fscale; fk.y *= fscale; }
                const bool dd = fabs(beta) > 10.0 * 2.220446049250313e-16;
                const double2 star = make_double2((fk.x - ak1 * prev.x) / beta, (fk.y - ak1 * prev.y) / beta);
                prev = dd ? star : ob[n];
                phi[q] = prev;
            }
        }
    }
    for (int k0 = Nz - 2; k0 >= 0; k0 -= TB) {
        double tb[TB];
        double2 pb[TB];
#pragma unroll
        for (int n = 0; n < TB; ++n) {
            const int k = k0 - n;
            if (k >= 0) { tb[n] = t[qb - (long)n * stb]; pb[n] = phi[q - (long)(n + 1) * st]; }     // t[k+1], phi[k]
        }
#pragma unroll
        for (int n = 0; n < TB; ++n) {
            const int k = k0 - n;
            if (k >= 0) {
                q -= st; qb -= stb;
                double2 cur = pb[n];
                cur.x -= tb[n] * prev.x;
                cur.y -= tb[n] * prev.y;
                phi[q] = cur;
                prev = cur;
            }
        }
    }
}

// `ϕ .= ϕ .- mean(ϕ)` (fourier_tridiagonal_poisson_solver.jl:233) applied in spectral space: the volume mean lives in
// the (kx, ky) = (0, 0) column, mean(ϕ) * Nx*Ny = mean_k ϕ̂(0,0,k). One workgroup.
__global__ void __launch_bounds__(256) remove_mean_mode_kernel(double2 *phi, long plane_stride, int Nz) {
    __shared__ double sx[256], sy[256];
    double ax = 0, ay = 0;
    for (int k = threadIdx.x; k < Nz; k += 256) { ax += phi[k * plane_stride].x; ay += phi[k * plane_stride].y; }
    sx[threadIdx.x] = ax; sy[threadIdx.x] = ay;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sx[threadIdx.x] += sx[threadIdx.x + s]; sy[threadIdx.x] += sy[threadIdx.x + s]; }
        __syncthreads();
    }
    const double mx = sx[0] / (double)Nz, my = sy[0] / (double)Nz;
    for (int k = threadIdx.x; k < Nz; k += 256) { phi[k * plane_stride].x -= mx; phi[k * plane_stride].y -= my; }
}

// dense real (Nx, Ny, Nz) -> interior of a haloed (C,C,C) field (fallback when the strided C2R plan is unavailable)
__global__ void __launch_bounds__(256) copy_dense_real_kernel(DGrid g, FView phi, const double *src) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    phi.at(i, j, k) = src[(long)(i - 1) + (long)g.Nx * ((j - 1) + (long)g.Ny * (k - 1))];
}

// ---------------------------------------------------------------------------------------------------------------------
// Irregular x partitions (remainder columns on the last rank, distributed_grids.jl:44-58): the gathered pressure solve. Every rank
// contributes its source term with rows padded to the widest slab (nmax columns); assemble the global dense right-hand side from
// the R gathered pieces, and -- after the single-GPU solver ran on it -- take the own slab back out of the global solution.
// first[q] = global index (0-based) of rank q's first column, first[R] = Nx_global.
// ---------------------------------------------------------------------------------------------------------------------
// With a pencil (Rx x Ry) partition the pieces are (nxmax, nymax, Nz) blocks, rank = ix * Ry + iy (index2rank, distributed_architectures.jl:354).
#define OCN_MAX_RANKS 64
struct SlabTable { int first[OCN_MAX_RANKS + 1]; int R; int firsty[OCN_MAX_RANKS + 1]; int Ry; };
template <bool COMPLEX>
__global__ void __launch_bounds__(256) gather_assemble_kernel(const double *all, void *dst, SlabTable t, int nmax, int nymax, int Nxg, int Nyg, int Nz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = blockIdx.z;
    if (i >= Nxg || j >= Nyg || k >= Nz) return;
    int q = 0, qy = 0;
    while (q + 1 < t.R && i >= t.first[q + 1]) ++q;
    while (qy + 1 < t.Ry && j >= t.firsty[qy + 1]) ++qy;
    const size_t r = (size_t)q * t.Ry + qy;
    const double v = all[(size_t)(i - t.first[q]) + (size_t)nmax * ((j - t.firsty[qy]) + (size_t)nymax * (k + (size_t)Nz * r))];
    const size_t d = (size_t)i + (size_t)Nxg * (j + (size_t)Nyg * k);
    if (COMPLEX) ((double2 *)dst)[d] = make_double2(v, 0.0);
    else ((double *)dst)[d] = v;
}
// local haloed phi(i, j, k) = global haloed gphi(off + i, offy + j, k) over the local interior
__global__ void __launch_bounds__(256) slab_extract_kernel(DGrid g, FView phi, FView gphi, int off, int offy) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    phi.at(i, j, k) = gphi.at(off + i, offy + j, k);
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-dimension transforms for grids with Bounded (cosine-transform) directions -- DiscreteTransform on the GPU
// (Solvers/discrete_transforms.jl:108-175, index_permutations.jl:18-72, plan_transforms.jl:39-136).
// Each direction d is transformed on its own: GATHER the lines of A (Nx, Ny, Nz) along d into a line-contiguous buffer B
// (B[m + N*line]), run ONE unit-stride batched complex FFT, SCATTER back. The cosine transforms ride on the same FFT
// (Makhoul 1980): forward  v = even-odd permutation of x, V = FFT(v), Y[k] = 2 Re(e^{-iπk/2N} V[k])   (= FFTW REDFT10);
// backward V[k] = e^{iπk/2N} (Y[k] - i Y[N-k]) / 2, v = N * IFFT(V), x = unpermuted v  -- i.e. N x like an unnormalised
// inverse FFT, so every direction contributes the same 1/N to the normalisation. Bounded directions are transformed FIRST
// on the way in and LAST on the way out (plan_transforms.jl:44-65) so that their input is real.
// mode 0: Periodic (plain copy); 1: Bounded forward; 2: Bounded backward.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int line_perm(int n, int N) { return (n & 1) ? N - 1 - (n >> 1) : (n >> 1); }

__global__ void __launch_bounds__(256) line_gather_kernel(const double2 *A, double2 *B, int Nx, int Ny, int Nz, int d, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = blockIdx.z;
    if (i >= Nx || j >= Ny || k >= Nz) return;
    const long sx = 1, sy = Nx, sz = (long)Nx * Ny;
    const long q = i * sx + j * sy + k * sz;
    const int n = d == 0 ? i : (d == 1 ? j : k);
    const int N = d == 0 ? Nx : (d == 1 ? Ny : Nz);
    const long sd = d == 0 ? sx : (d == 1 ? sy : sz);
    const long line = d == 0 ? (j + (long)Ny * k) : (d == 1 ? (i + (long)Nx * k) : (i + (long)Nx * j));
    double2 val = A[q];
    int m = n;
    if (mode == 1) {
        m = line_perm(n, N);
        val.y = 0.0;
    } else if (mode == 2) {
        const double yk = val.x;
        const double ynk = n == 0 ? 0.0 : A[q + (long)(N - 2 * n) * sd].x;     // Y[N - n], Y[N] := 0
        double sn, cs;
        sincospi((double)n / (2.0 * (double)N), &sn, &cs);
        // e^{iπn/2N} (yk - i ynk) / 2
        val = make_double2(0.5 * (cs * yk + sn * ynk), 0.5 * (sn * yk - cs * ynk));
    }
    B[m + (long)N * line] = val;
}

__global__ void __launch_bounds__(256) line_scatter_kernel(const double2 *B, double2 *A, int Nx, int Ny, int Nz, int d, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = blockIdx.z;
    if (i >= Nx || j >= Ny || k >= Nz) return;
    const long q = i + (long)Nx * (j + (long)Ny * k);
    const int n = d == 0 ? i : (d == 1 ? j : k);
    const int N = d == 0 ? Nx : (d == 1 ? Ny : Nz);
    const long line = d == 0 ? (j + (long)Ny * k) : (d == 1 ? (i + (long)Nx * k) : (i + (long)Nx * j));
    if (mode == 0) {
        A[q] = B[n + (long)N * line];
    } else if (mode == 1) {
        const double2 V = B[n + (long)N * line];
        double sn, cs;
        sincospi((double)n / (2.0 * (double)N), &sn, &cs);
        A[q] = make_double2(2.0 * (cs * V.x + sn * V.y), 0.0);                   // 2 Re(e^{-iπn/2N} V)
    } else {
        A[q] = make_double2(B[line_perm(n, N) + (long)N * line].x, 0.0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused z pass of the FFT-based Poisson solve on the Hermitian half spectrum (Nxs, Ny, Nz), z Periodic, Nz a power of two:
// forward FFT along z, ϕ̂ = -b̂ / (λx + λy + λz) (fft_based_poisson_solver.jl:108-118), inverse FFT along z -- ONE pass over the
// array instead of three (rocFFT column transform, divide kernel, rocFFT column transform). A workgroup holds ZL z-lines
// (consecutive in x: 128-B rows) in LDS. Forward = radix-2 decimation in frequency (natural -> bit-reversed order), the divide
// runs in bit-reversed order, inverse = radix-2 decimation in time (bit-reversed -> natural): in place, no reordering pass.
// tw[m] = exp(-2πi m / Nz), m < Nz/2 (host-computed). `scale` folds the normalisation of the whole 3-D inverse transform.
// ---------------------------------------------------------------------------------------------------------------------
template <int ZL>
__global__ void __launch_bounds__(256) zline_solve_kernel(double2 *hc, const double2 *tw, const double *lx, const double *ly,
                                                          const double *lz, int Nxs, int Ny, int Nz, int logn, double scale, int pitch = 0) {
    extern __shared__ double2 zbuf[];                 // [Nz][ZL]
    const int il = threadIdx.x % ZL, kq = threadIdx.x / ZL;       // 32 k-rows per pass
    const int i0 = blockIdx.x * ZL, j = blockIdx.y;
    const int i = i0 + il;
    const bool live = i < Nxs;
    const int ldx = pitch > 0 ? pitch : Nxs;          // row pitch of the spectrum (>= Nxs: padded to whole 128-B rows on the split path)
    const long plane = (long)ldx * Ny, base = (long)i + (long)ldx * j;
    for (int k = kq; k < Nz; k += 256 / ZL) zbuf[k * ZL + il] = live ? hc[base + plane * k] : make_double2(0.0, 0.0);
    __syncthreads();
    const int half = Nz >> 1, quarter = Nz >> 2, KQ = 256 / ZL;
#define ZB(n) zbuf[(n) * ZL + il]
#define CMUL(ar, ai, w) make_double2((ar) * (w).x - (ai) * (w).y, (ar) * (w).y + (ai) * (w).x)        /* (ar + i ai) * w       */
#define CMULC(v, w) make_double2((v).x * (w).x + (v).y * (w).y, (v).y * (w).x - (v).x * (w).y)       /* v * conj(w)           */
    // forward, decimation in frequency (natural -> bit-reversed): spans Nz/2, Nz/4, ..., 1; two radix-2 stages are fused in
    // registers (radix-4 butterflies: half the LDS traffic and barriers), preceded by one radix-2 stage when log2 Nz is odd
    int h = half, st = 1;
    if (logn & 1) {
        for (int q = kq; q < half; q += KQ) {
            const int jj = q & (h - 1), a = ((q - jj) << 1) + jj, b = a + h;
            const double2 xa = ZB(a), xb = ZB(b), w = tw[jj * st];
            ZB(a) = make_double2(xa.x + xb.x, xa.y + xb.y);
            ZB(b) = CMUL(xa.x - xb.x, xa.y - xb.y, w);
        }
        __syncthreads();
        h >>= 1; st <<= 1;
    }
    for (; h >= 2; h >>= 2, st <<= 2) {
        const int h2 = h >> 1;
        for (int q = kq; q < quarter; q += KQ) {
            const int jj = q & (h2 - 1), a = ((q - jj) << 2) + jj;             // blocks of 2h elements, jj < h/2
            const double2 x0 = ZB(a), x1 = ZB(a + h2), x2 = ZB(a + h), x3 = ZB(a + h + h2);
            const double2 wa = tw[jj * st], wb = tw[(jj + h2) * st], wc = tw[jj * 2 * st];
            const double2 y0 = make_double2(x0.x + x2.x, x0.y + x2.y), y2 = CMUL(x0.x - x2.x, x0.y - x2.y, wa);
            const double2 y1 = make_double2(x1.x + x3.x, x1.y + x3.y), y3 = CMUL(x1.x - x3.x, x1.y - x3.y, wb);
            ZB(a) = make_double2(y0.x + y1.x, y0.y + y1.y);
            ZB(a + h2) = CMUL(y0.x - y1.x, y0.y - y1.y, wc);
            ZB(a + h) = make_double2(y2.x + y3.x, y2.y + y3.y);
            ZB(a + h + h2) = CMUL(y2.x - y3.x, y2.y - y3.y, wc);
        }
        __syncthreads();
    }
    // spectral divide in bit-reversed order
    if (live) {
        const double lxy = lx[i] + ly[j];
        for (int p = kq; p < Nz; p += KQ) {
            const int k = (int)(__brev((unsigned)p) >> (32 - logn));
            double2 v = ZB(p);
            const double lam = lxy + lz[k] - 0.0;
            v.x = -(v.x * scale) / lam;
            v.y = -(v.y * scale) / lam;
            if (i == 0 && j == 0 && k == 0) v = make_double2(0.0, 0.0);
            ZB(p) = v;
        }
    }
    __syncthreads();
    // inverse, decimation in time (bit-reversed -> natural): spans 1, 2, ..., Nz/2, conjugate twiddles, fused in pairs
    h = 1; st = half;
    for (; (h << 1) <= half; h <<= 2, st >>= 2) {
        for (int q = kq; q < quarter; q += KQ) {
            const int jj = q & (h - 1), a = ((q - jj) << 2) + jj;              // blocks of 4h elements, jj < h
            const double2 x0 = ZB(a), x1 = ZB(a + h), x2 = ZB(a + 2 * h), x3 = ZB(a + 3 * h);
            const double2 wa = tw[jj * st], wb = tw[jj * (st >> 1)], wc = tw[(jj + h) * (st >> 1)];
            const double2 t1 = CMULC(x1, wa), t3 = CMULC(x3, wa);
            const double2 y0 = make_double2(x0.x + t1.x, x0.y + t1.y), y1 = make_double2(x0.x - t1.x, x0.y - t1.y);
            const double2 y2 = make_double2(x2.x + t3.x, x2.y + t3.y), y3 = make_double2(x2.x - t3.x, x2.y - t3.y);
            const double2 u2 = CMULC(y2, wb), u3 = CMULC(y3, wc);
            ZB(a) = make_double2(y0.x + u2.x, y0.y + u2.y);
            ZB(a + 2 * h) = make_double2(y0.x - u2.x, y0.y - u2.y);
            ZB(a + h) = make_double2(y1.x + u3.x, y1.y + u3.y);
            ZB(a + 3 * h) = make_double2(y1.x - u3.x, y1.y - u3.y);
        }
        __syncthreads();
    }
    if (h <= half) {                          // one radix-2 stage left when log2 Nz is odd (h == half here)
        for (int q = kq; q < half; q += KQ) {
            const int jj = q & (h - 1), a = ((q - jj) << 1) + jj, b = a + h;
            const double2 xa = ZB(a), t = CMULC(ZB(b), tw[jj * st]);
            ZB(a) = make_double2(xa.x + t.x, xa.y + t.y);
            ZB(b) = make_double2(xa.x - t.x, xa.y - t.y);
        }
        __syncthreads();
    }
#undef ZB
#undef CMUL
#undef CMULC
    if (live)
        for (int k = kq; k < Nz; k += 256 / ZL) hc[base + plane * k] = zbuf[k * ZL + il];
}

// One-directional FFT of length N = 2^logn along the SLOWEST dimension of a (C, N) complex array (element (c, n) at c + C*n): the
// column transform rocFFT runs with its row kernel when it is planned as a 1-D strided transform (measured 129 us for 67 MB; this
// kernel: the same LDS machinery as zline_solve_kernel, ZL consecutive lines per workgroup: 8 = 128-B rows; 4 for lines of 512 and more,
// whose 64 KB of LDS per 8 lines leave two workgroups per CU -- measured at 512^3: 54.0 -> 51.9 ms/step with 4, 54.1 with 2; at 256^3 8 wins).
// forward: natural -> radix-4 DIF -> stored through the bit-reversal; inverse: loaded through the bit-reversal -> DIT -> natural,
// times `scale`. Same arithmetic as rocFFT's to round-off, not bitwise.
template <int ZL>
__global__ void __launch_bounds__(256) strided_line_fft_kernel(double2 *data, const double2 *tw, long C, int N, int logn, int inverse,
                                                               double scale, long plane_stride = 0) {
    extern __shared__ double2 zbuf[];                 // [N][ZL]
    data += plane_stride * blockIdx.y;                // gridDim.y independent (C, N) arrays `plane_stride` elements apart
    const int il = threadIdx.x % ZL, kq = threadIdx.x / ZL;
    const long c = (long)blockIdx.x * ZL + il;
    const bool live = c < C;
    const int half = N >> 1, quarter = N >> 2, KQ = 256 / ZL;
#define ZB(n) zbuf[(n) * ZL + il]
#define CMUL(ar, ai, w) make_double2((ar) * (w).x - (ai) * (w).y, (ar) * (w).y + (ai) * (w).x)
#define CMULC(v, w) make_double2((v).x * (w).x + (v).y * (w).y, (v).y * (w).x - (v).x * (w).y)
    if (!inverse) {
        for (int k = kq; k < N; k += KQ) zbuf[k * ZL + il] = live ? data[c + C * k] : make_double2(0.0, 0.0);
        __syncthreads();
        int h = half, st = 1;
        if (logn & 1) {
            for (int q = kq; q < half; q += KQ) {
                const int jj = q & (h - 1), a = ((q - jj) << 1) + jj, b = a + h;
                const double2 xa = ZB(a), xb = ZB(b), w = tw[jj * st];
                ZB(a) = make_double2(xa.x + xb.x, xa.y + xb.y);
                ZB(b) = CMUL(xa.x - xb.x, xa.y - xb.y, w);
            }
            __syncthreads();
            h >>= 1; st <<= 1;
        }
        for (; h >= 2; h >>= 2, st <<= 2) {
            const int h2 = h >> 1;
            for (int q = kq; q < quarter; q += KQ) {
                const int jj = q & (h2 - 1), a = ((q - jj) << 2) + jj;
                const double2 x0 = ZB(a), x1 = ZB(a + h2), x2 = ZB(a + h), x3 = ZB(a + h + h2);
                const double2 wa = tw[jj * st], wb = tw[(jj + h2) * st], wc = tw[jj * 2 * st];
                const double2 y0 = make_double2(x0.x + x2.x, x0.y + x2.y), y2 = CMUL(x0.x - x2.x, x0.y - x2.y, wa);
                const double2 y1 = make_double2(x1.x + x3.x, x1.y + x3.y), y3 = CMUL(x1.x - x3.x, x1.y - x3.y, wb);
                ZB(a) = make_double2(y0.x + y1.x, y0.y + y1.y);
                ZB(a + h2) = CMUL(y0.x - y1.x, y0.y - y1.y, wc);
                ZB(a + h) = make_double2(y2.x + y3.x, y2.y + y3.y);
                ZB(a + h + h2) = CMUL(y2.x - y3.x, y2.y - y3.y, wc);
            }
            __syncthreads();
        }
        if (live)
            for (int p = kq; p < N; p += KQ) data[c + C * (long)(__brev((unsigned)p) >> (32 - logn))] = zbuf[p * ZL + il];
    } else {
        for (int k = kq; k < N; k += KQ)
            zbuf[(int)(__brev((unsigned)k) >> (32 - logn)) * ZL + il] = live ? data[c + C * k] : make_double2(0.0, 0.0);
        __syncthreads();
        int h = 1, st = half;
        for (; (h << 1) <= half; h <<= 2, st >>= 2) {
            for (int q = kq; q < quarter; q += KQ) {
                const int jj = q & (h - 1), a = ((q - jj) << 2) + jj;
                const double2 x0 = ZB(a), x1 = ZB(a + h), x2 = ZB(a + 2 * h), x3 = ZB(a + 3 * h);
                const double2 wa = tw[jj * st], wb = tw[jj * (st >> 1)], wc = tw[(jj + h) * (st >> 1)];
                const double2 t1 = CMULC(x1, wa), t3 = CMULC(x3, wa);
                const double2 y0 = make_double2(x0.x + t1.x, x0.y + t1.y), y1 = make_double2(x0.x - t1.x, x0.y - t1.y);
                const double2 y2 = make_double2(x2.x + t3.x, x2.y + t3.y), y3 = make_double2(x2.x - t3.x, x2.y - t3.y);
                const double2 u2 = CMULC(y2, wb), u3 = CMULC(y3, wc);
                ZB(a) = make_double2(y0.x + u2.x, y0.y + u2.y);
                ZB(a + 2 * h) = make_double2(y0.x - u2.x, y0.y - u2.y);
                ZB(a + h) = make_double2(y1.x + u3.x, y1.y + u3.y);
                ZB(a + 3 * h) = make_double2(y1.x - u3.x, y1.y - u3.y);
            }
            __syncthreads();
        }
        if (h <= half) {
            for (int q = kq; q < half; q += KQ) {
                const int jj = q & (h - 1), a = ((q - jj) << 1) + jj, b = a + h;
                const double2 xa = ZB(a), t = CMULC(ZB(b), tw[jj * st]);
                ZB(a) = make_double2(xa.x + t.x, xa.y + t.y);
                ZB(b) = make_double2(xa.x - t.x, xa.y - t.y);
            }
            __syncthreads();
        }
        if (live)
            for (int k = kq; k < N; k += KQ) {
                const double2 v = zbuf[k * ZL + il];
                data[c + C * k] = make_double2(v.x * scale, v.y * scale);
            }
    }
#undef ZB
#undef CMUL
#undef CMULC
}

// The x stage of the distributed FFT solve in ONE pass (distributed_fft_based_poisson_solver.jl:152-166 with the transposes'
// unpack / pack folded in): a workgroup gathers L whole x-lines (length Nxg = R * Nxl, a power of two) from the all-to-all
// receive buffer -- chunk p holds columns [p*Nxl, (p+1)*Nxl) of every line (il fastest) --, transforms them forward (radix-4
// DIF), divides by the eigenvalues in bit-reversed order, transforms back (DIT) and scatters them into the send buffer in
// the same chunk layout. Replaces unpack + rocFFT + divide + rocFFT + pack (5 passes). LDS: L * Nxg complex.
__global__ void __launch_bounds__(256) xline_solve_kernel(const double2 *recv, double2 *send, const double2 *tw, const double *lx,
                                                          const double *ly, const double *lz, int R, int Nxl, int Nyc, int Nz,
                                                          int logn, int L, int joff, int Ny, double scale) {
    extern __shared__ double2 zbuf[];                 // [L][Nxg]
    const int N = R * Nxl;
    const long nlines = (long)Nyc * Nz, line0 = (long)blockIdx.x * L;
    const long chunk = (long)Nxl * nlines;
    const int total = L * N;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int ln = e >> logn, n = e & (N - 1);
        const long line = line0 + ln;
        const int pch = n / Nxl, il = n - pch * Nxl;
        zbuf[e] = line < nlines ? recv[pch * chunk + il + (long)Nxl * line] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int half = N >> 1, quarter = N >> 2;
#define ZB(n) zbuf[lo + (n)]
#define CMUL(ar, ai, w) make_double2((ar) * (w).x - (ai) * (w).y, (ar) * (w).y + (ai) * (w).x)
#define CMULC(v, w) make_double2((v).x * (w).x + (v).y * (w).y, (v).y * (w).x - (v).x * (w).y)
    int h = half, st = 1;
    if (logn & 1) {
        for (int e = threadIdx.x; e < L * half; e += 256) {
            const int ln = e >> (logn - 1), q = e & (half - 1), lo = ln << logn;
            const int jj = q & (h - 1), a = ((q - jj) << 1) + jj, b = a + h;
            const double2 xa = ZB(a), xb = ZB(b), w = tw[jj * st];
            ZB(a) = make_double2(xa.x + xb.x, xa.y + xb.y);
            ZB(b) = CMUL(xa.x - xb.x, xa.y - xb.y, w);
        }
        __syncthreads();
        h >>= 1; st <<= 1;
    }
    for (; h >= 2; h >>= 2, st <<= 2) {
        const int h2 = h >> 1;
        for (int e = threadIdx.x; e < L * quarter; e += 256) {
            const int ln = e >> (logn - 2), q = e & (quarter - 1), lo = ln << logn;
            const int jj = q & (h2 - 1), a = ((q - jj) << 2) + jj;
            const double2 x0 = ZB(a), x1 = ZB(a + h2), x2 = ZB(a + h), x3 = ZB(a + h + h2);
            const double2 wa = tw[jj * st], wb = tw[(jj + h2) * st], wc = tw[jj * 2 * st];
            const double2 y0 = make_double2(x0.x + x2.x, x0.y + x2.y), y2 = CMUL(x0.x - x2.x, x0.y - x2.y, wa);
            const double2 y1 = make_double2(x1.x + x3.x, x1.y + x3.y), y3 = CMUL(x1.x - x3.x, x1.y - x3.y, wb);
            ZB(a) = make_double2(y0.x + y1.x, y0.y + y1.y);
            ZB(a + h2) = CMUL(y0.x - y1.x, y0.y - y1.y, wc);
            ZB(a + h) = make_double2(y2.x + y3.x, y2.y + y3.y);
            ZB(a + h + h2) = CMUL(y2.x - y3.x, y2.y - y3.y, wc);
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < total; e += 256) {
        const int ln = e >> logn, pz = e & (N - 1);
        const long line = line0 + ln;
        if (line >= nlines) continue;
        const int jl = (int)(line % Nyc), k = (int)(line / Nyc);
        const int i = (int)(__brev((unsigned)pz) >> (32 - logn));
        const int jg = min(joff + jl, Ny - 1);
        double2 v = zbuf[e];
        const double lam = (lx[i] + ly[jg]) + lz[k] - 0.0;
        v.x = -(v.x * scale) / lam;
        v.y = -(v.y * scale) / lam;
        if (i == 0 && joff + jl == 0 && k == 0) v = make_double2(0.0, 0.0);
        zbuf[e] = v;
    }
    __syncthreads();
    h = 1; st = half;
    for (; (h << 1) <= half; h <<= 2, st >>= 2) {
        for (int e = threadIdx.x; e < L * quarter; e += 256) {
            const int ln = e >> (logn - 2), q = e & (quarter - 1), lo = ln << logn;
            const int jj = q & (h - 1), a = ((q - jj) << 2) + jj;
            const double2 x0 = ZB(a), x1 = ZB(a + h), x2 = ZB(a + 2 * h), x3 = ZB(a + 3 * h);
            const double2 wa = tw[jj * st], wb = tw[jj * (st >> 1)], wc = tw[(jj + h) * (st >> 1)];
            const double2 t1 = CMULC(x1, wa), t3 = CMULC(x3, wa);
            const double2 y0 = make_double2(x0.x + t1.x, x0.y + t1.y), y1 = make_double2(x0.x - t1.x, x0.y - t1.y);
            const double2 y2 = make_double2(x2.x + t3.x, x2.y + t3.y), y3 = make_double2(x2.x - t3.x, x2.y - t3.y);
            const double2 u2 = CMULC(y2, wb), u3 = CMULC(y3, wc);
            ZB(a) = make_double2(y0.x + u2.x, y0.y + u2.y);
            ZB(a + 2 * h) = make_double2(y0.x - u2.x, y0.y - u2.y);
            ZB(a + h) = make_double2(y1.x + u3.x, y1.y + u3.y);
            ZB(a + 3 * h) = make_double2(y1.x - u3.x, y1.y - u3.y);
        }
        __syncthreads();
    }
    if (h <= half) {
        for (int e = threadIdx.x; e < L * half; e += 256) {
            const int ln = e >> (logn - 1), q = e & (half - 1), lo = ln << logn;
            const int jj = q & (h - 1), a = ((q - jj) << 1) + jj, b = a + h;
            const double2 xa = ZB(a), t = CMULC(ZB(b), tw[jj * st]);
            ZB(a) = make_double2(xa.x + t.x, xa.y + t.y);
            ZB(b) = make_double2(xa.x - t.x, xa.y - t.y);
        }
        __syncthreads();
    }
#undef ZB
#undef CMUL
#undef CMULC
    for (int e = threadIdx.x; e < total; e += 256) {
        const int ln = e >> logn, n = e & (N - 1);
        const long line = line0 + ln;
        if (line >= nlines) continue;
        const int pch = n / Nxl, il = n - pch * Nxl;
        send[pch * chunk + il + (long)Nxl * line] = zbuf[e];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Substructured x solve of the distributed FFT Poisson solver (z Periodic): after the local (y, z) transform every (ky, kz) mode
// is an independent PERIODIC CONSTANT-COEFFICIENT TRIDIAGONAL system along the partitioned x direction,
//     a (p[i-1] - 2 p[i] + p[i+1]) - (λy + λz) p[i] = r[i],   a = 1/Δx².
// Instead of transposing the whole spectrum twice (2 all-to-alls of ~140 MB per rank -- one xGMI link per peer) each rank solves
// its slab with Dirichlet ends (Thomas, real coefficients precomputed once), the ranks exchange TWO numbers per mode (first and
// last value: an all-gather of ~1 MB), the 2R interface unknowns per mode follow from a system that is block-circulant in the
// rank index (constant coefficients => diagonalised by an R-point DFT: 2x2 solves), and the slab solution is corrected with
// p = y - a gL s - a gR s[reversed], s = T⁻¹ e₀. The null mode (ky = kz = 0) is pinned and shifted to zero mean like the
// reference's ϕ̂[1,1,1] = 0. Same solution as the transposed FFT solve to round-off (tested at 1e-12).
// Spectral layout here: Y[m + M*i], m = ky + Nyh*kz fastest (coalesced sweeps along x), i = local x index.
// ---------------------------------------------------------------------------------------------------------------------
// setup, one thread per mode: rden[i] = 1 / (b - a cp[i-1]), cp[i] = a rden[i], s = T⁻¹ e₀ (T = tridiag(a, b, a), b = -2a - λ)
__global__ void __launch_bounds__(256) sub_setup_kernel(int M, int Nyh, int N, double a, const double *ly, const double *lz, double *rden,
                                                        double *cp, double *svec, double *ssum0) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int jy = m % Nyh, k = m / Nyh;
    const double b = -2.0 * a - (ly[jy] + lz[k]);
    double c = 0.0;
    for (int i = 0; i < N; ++i) {
        const double rd = 1.0 / (b - a * c);
        c = a * rd;
        rden[m + (long)M * i] = rd;
        cp[m + (long)M * i] = c;
    }
    double y = rden[m];                       // forward sweep of e₀: y[0] = 1/den[0], y[i] = -a y[i-1] / den[i]
    svec[m] = y;
    for (int i = 1; i < N; ++i) { y = (-a * y) * rden[m + (long)M * i]; svec[m + (long)M * i] = y; }
    for (int i = N - 2; i >= 0; --i) svec[m + (long)M * i] -= cp[m + (long)M * i] * svec[m + (long)M * (i + 1)];
    if (m == 0) {                             // Σ_i s[i] of the null mode (its global mean needs it at every solve)
        double t = 0.0;
        for (int i = 0; i < N; ++i) t += svec[(long)M * i];
        *ssum0 = t;
    }
}

// separate the paired columns (see dist_pack_forward_kernel) and transpose (ih, k, j) -> (m, i) through an LDS tile
__global__ void __launch_bounds__(256) sub_separate_kernel(const double2 *Z, double2 *Y, int Nxl, int Nxh, int Ny, int Nyh, int Nz) {
    __shared__ double2 tA[16][17], tB[16][17];
    const int k = blockIdx.z, tx = threadIdx.x, ty = threadIdx.y;
    const long M = (long)Nyh * Nz;
    {
        const int ih = blockIdx.x * 16 + tx, j = blockIdx.y * 16 + ty;
        if (ih < Nxh && j < Nyh) {
            const int jm = j == 0 ? 0 : Ny - j, km = k == 0 ? 0 : Nz - k;
            const double2 z1 = Z[ih + (long)Nxh * (k + (long)Nz * j)], z2 = Z[ih + (long)Nxh * (km + (long)Nz * jm)];
            tA[ty][tx] = make_double2(0.5 * (z1.x + z2.x), 0.5 * (z1.y - z2.y));
            tB[ty][tx] = make_double2(0.5 * (z1.y + z2.y), 0.5 * (z2.x - z1.x));
        }
    }
    __syncthreads();
    {
        const int j = blockIdx.y * 16 + tx, ih = blockIdx.x * 16 + ty;
        if (ih < Nxh && j < Nyh) {
            const long m = j + (long)Nyh * k;
            Y[m + M * (2 * ih)] = tA[tx][ty];
            if (2 * ih + 1 < Nxl) Y[m + M * (2 * ih + 1)] = tB[tx][ty];
        }
    }
}

// Thomas sweeps along x for every mode (in place), then the payload: first / last value per mode and, for mode 0, the sum
// ZF: the spectrum is stored z-fastest, Y[kz + Nzp*(i + N*ky)] (row pitch Nzp >= Nzh) with m = kz + Nzh*ky (what R2C along z + a strided y transform leave
// behind); otherwise mode-fastest, Y[m + M*i]. The factor arrays are mode-fastest in both cases.
template <bool ZF>
__global__ void __launch_bounds__(64) sub_thomas_kernel(long M, int N, double a, const double *__restrict__ rden, const double *__restrict__ cp,
                                                        double2 *__restrict__ Yin, double2 *__restrict__ payload, int Nzh, int Nzp) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    // element (m, i) of the spectrum: base + stride * i (ZF: rows of Nzh modes stored with pitch Nzp >= Nzh, whole 128-B lines)
    const long ybase = ZF ? (m % Nzh) + (long)Nzp * N * (m / Nzh) : m;
    const long ystride = ZF ? (long)Nzp : M;
    double2 *__restrict__ Y = Yin + ybase - m;          // so that Y[m + ystride * i] below is the element
    constexpr int TB = 8;
    double2 prev = make_double2(0.0, 0.0);
    for (int i0 = 0; i0 < N; i0 += TB) {
        double2 r[TB];
        double d[TB];
#pragma unroll
        for (int n = 0; n < TB; ++n)
            if (i0 + n < N) { r[n] = Y[m + ystride * (i0 + n)]; d[n] = rden[m + M * (i0 + n)]; }
#pragma unroll
        for (int n = 0; n < TB; ++n)
            if (i0 + n < N) {
                prev = make_double2((r[n].x - a * prev.x) * d[n], (r[n].y - a * prev.y) * d[n]);
                Y[m + ystride * (i0 + n)] = prev;
            }
    }
    double2 sum = prev;                           // prev = y[N-1] is final already
    const double2 last = prev;
    for (int i0 = N - 2; i0 >= 0; i0 -= TB) {
        double2 y[TB];
        double c[TB];
#pragma unroll
        for (int n = 0; n < TB; ++n)
            if (i0 - n >= 0) { y[n] = Y[m + ystride * (i0 - n)]; c[n] = cp[m + M * (i0 - n)]; }
#pragma unroll
        for (int n = 0; n < TB; ++n)
            if (i0 - n >= 0) {
                prev = make_double2(y[n].x - c[n] * prev.x, y[n].y - c[n] * prev.y);
                Y[m + ystride * (i0 - n)] = prev;
                sum.x += prev.x; sum.y += prev.y;
            }
    }
    payload[m] = prev;                            // y[0]
    payload[M + m] = last;                        // y[N-1]
    if (m == 0) payload[2 * M] = sum;
}

// interface unknowns from the gathered payloads (R ranks x (2M + 1)): per mode an R-point DFT over the rank index, 2x2 solves, and
// the two values this rank needs: gL = l[rank-1], gR = f[rank+1]. out[m] = gL, out[M+m] = gR, out[2M] = mean of mode 0
__global__ void __launch_bounds__(256) sub_interface_kernel(long M, int Nyh, int N, int R, int rank, double a, const double *ly,
                                                            const double *lz, const double *s_first, const double *s_last, const double *ssum0,
                                                            const double2 *gathered, double2 *out) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const long stride = 2 * M + 1;
    const double alpha = a * s_first[m], beta = a * s_last[m];          // s = T⁻¹e₀: its first and its last entry
    const bool null_mode = m == 0 && (ly[0] + lz[0]) == 0.0;
    double2 gl = make_double2(0.0, 0.0), gr = gl;
    double2 mean = gl;
    const int rl = (rank + R - 1) % R, rr = (rank + 1) % R;
    for (int q = 0; q < R; ++q) {
        double2 yF = make_double2(0.0, 0.0), yL = yF;           // DFT over the rank index: Σ_r y_r e^{-2πi q r / R}
        for (int r = 0; r < R; ++r) {
            double sn, cs;
            sincospi(-2.0 * (double)((q * r) % R) / (double)R, &sn, &cs);
            const double2 f = gathered[r * stride + m], l = gathered[r * stride + M + m];
            yF.x += f.x * cs - f.y * sn; yF.y += f.x * sn + f.y * cs;
            yL.x += l.x * cs - l.y * sn; yL.y += l.x * sn + l.y * cs;
        }
        double so, co;
        sincospi(2.0 * (double)q / (double)R, &so, &co);        // ω = e^{2πi q / R}
        // [1 + β ω,  α ω⁻¹; α ω,  1 + β ω⁻¹] [f̂; l̂] = [ŷF; ŷL]
        const double2 A11 = make_double2(1.0 + beta * co, beta * so), A22 = make_double2(1.0 + beta * co, -beta * so);
        const double2 A12 = make_double2(alpha * co, -alpha * so), A21 = make_double2(alpha * co, alpha * so);
        double2 fh, lh;
        if (null_mode && q == 0) {                               // singular: pin l̂ = 0 (compatibility: Σ rhs = 0), shift to zero mean below
            const double d2 = A11.x * A11.x + A11.y * A11.y;
            fh = make_double2((yF.x * A11.x + yF.y * A11.y) / d2, (yF.y * A11.x - yF.x * A11.y) / d2);
            lh = make_double2(0.0, 0.0);
        } else {
            const double2 det = make_double2((A11.x * A22.x - A11.y * A22.y) - (A12.x * A21.x - A12.y * A21.y),
                                             (A11.x * A22.y + A11.y * A22.x) - (A12.x * A21.y + A12.y * A21.x));
            const double d2 = det.x * det.x + det.y * det.y;
            const double2 nf = make_double2((A22.x * yF.x - A22.y * yF.y) - (A12.x * yL.x - A12.y * yL.y),
                                            (A22.x * yF.y + A22.y * yF.x) - (A12.x * yL.y + A12.y * yL.x));
            const double2 nl = make_double2((A11.x * yL.x - A11.y * yL.y) - (A21.x * yF.x - A21.y * yF.y),
                                            (A11.x * yL.y + A11.y * yL.x) - (A21.x * yF.y + A21.y * yF.x));
            fh = make_double2((nf.x * det.x + nf.y * det.y) / d2, (nf.y * det.x - nf.x * det.y) / d2);
            lh = make_double2((nl.x * det.x + nl.y * det.y) / d2, (nl.y * det.x - nl.x * det.y) / d2);
        }
        // inverse DFT at the two rank indices this rank needs
        double s1, c1, s2, c2;
        sincospi(2.0 * (double)((q * rl) % R) / (double)R, &s1, &c1);
        sincospi(2.0 * (double)((q * rr) % R) / (double)R, &s2, &c2);
        gl.x += (lh.x * c1 - lh.y * s1) / R; gl.y += (lh.x * s1 + lh.y * c1) / R;
        gr.x += (fh.x * c2 - fh.y * s2) / R; gr.y += (fh.x * s2 + fh.y * c2) / R;
        if (null_mode) {
            // Σ_r (gL_r + gR_r) = R (l̂_0 + f̂_0): only q = 0 survives the sum over ranks
            if (q == 0) { mean.x = fh.x + lh.x; mean.y = fh.y + lh.y; }
        }
    }
    out[m] = gl;
    out[M + m] = gr;
    if (m == 0) {
        double2 mu = make_double2(0.0, 0.0);
        if (null_mode) {
            // global sum of p = Σ_r Σ_i y_r[i] - a (Σ_r gL_r + Σ_r gR_r) Σ_i s[i];  Σ_r gL_r = l̂_0, Σ_r gR_r = f̂_0 (q = 0 of the DFT)
            double2 ysum = make_double2(0.0, 0.0);
            for (int r = 0; r < R; ++r) { ysum.x += gathered[r * stride + 2 * M].x; ysum.y += gathered[r * stride + 2 * M].y; }
            const double ssum = *ssum0;
            const double cnt = (double)R * (double)N;
            mu = make_double2((ysum.x - a * mean.x * ssum) / cnt, (ysum.y - a * mean.y * ssum) / cnt);
        }
        out[2 * M] = mu;
    }
}

// p = (y - a gL s - a gR s[N-1-i] - mean[mode 0]) * scale, transposed back to the paired (ih, k, j) layout with the upper half of
// the spectrum rebuilt from the conjugate symmetry (see dist_combine_backward_kernel)
__global__ void __launch_bounds__(256) sub_correct_combine_kernel(const double2 *Y, const double *svec, const double2 *iface, double2 *Z,
                                                                  int Nxl, int Nxh, int Ny, int Nyh, int Nz, double a, double scale) {
    __shared__ double2 tA[16][17], tB[16][17];
    const int k = blockIdx.z, tx = threadIdx.x, ty = threadIdx.y;
    const long M = (long)Nyh * Nz;
    {
        const int j = blockIdx.y * 16 + tx, ih = blockIdx.x * 16 + ty;
        if (ih < Nxh && j < Nyh) {
            const long m = j + (long)Nyh * k;
            const double2 gl = iface[m], gr = iface[M + m];
            const double2 mu = m == 0 ? iface[2 * M] : make_double2(0.0, 0.0);
            double2 v[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int i = 2 * ih + o;
                if (i < Nxl) {
                    const double2 y = Y[m + M * i];
                    const double s0 = svec[m + M * i], s1 = svec[m + M * (Nxl - 1 - i)];
                    v[o] = make_double2((y.x - a * gl.x * s0 - a * gr.x * s1 - mu.x) * scale, (y.y - a * gl.y * s0 - a * gr.y * s1 - mu.y) * scale);
                } else v[o] = make_double2(0.0, 0.0);
            }
            tA[tx][ty] = v[0];
            tB[tx][ty] = v[1];
        }
    }
    __syncthreads();
    {
        const int ih = blockIdx.x * 16 + tx, j = blockIdx.y * 16 + ty;
        if (ih < Nxh && j < Nyh) {
            const double2 A = tA[ty][tx], B = tB[ty][tx];
            Z[ih + (long)Nxh * (k + (long)Nz * j)] = make_double2(A.x - B.y, A.y + B.x);
            const int jm = Ny - j, km = k == 0 ? 0 : Nz - k;
            if (j != 0 && jm >= Nyh) Z[ih + (long)Nxh * (km + (long)Nz * jm)] = make_double2(A.x + B.y, B.x - A.y);
        }
    }
}

// z-fastest variant of the local stage (R2C along z, then y): p = (y - a gL s - a gR s[N-1-i] - mean[mode 0]) * scale, in place on
// Y[kz + Nzh*(i + N*ky)]; one thread per element, mode m = kz + Nzh*ky
__global__ void __launch_bounds__(256) sub_correct_zfast_kernel(double2 *Y, const double *svec, const double2 *iface, long M, int N, int Nzh,
                                                                int Ny, double a, double scale, int Nzp) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;          // index into the padded array (pitch Nzp)
    if (t >= (long)Nzp * N * Ny) return;
    const int kz = t % Nzp;
    if (kz >= Nzh) return;                                                 // padding
    const long r = t / Nzp;
    const int i = r % N, ky = r / N;
    const long m = kz + (long)Nzh * ky;
    const double2 gl = iface[m], gr = iface[M + m];
    const double2 mu = m == 0 ? iface[2 * M] : make_double2(0.0, 0.0);
    const double2 y = Y[t];
    const double s0 = svec[m + M * i], s1 = svec[m + M * (N - 1 - i)];
    Y[t] = make_double2((y.x - a * gl.x * s0 - a * gr.x * s1 - mu.x) * scale, (y.y - a * gl.y * s0 - a * gr.y * s1 - mu.y) * scale);
    (void)Ny;
}

// compute_source_term! into a dense real array stored z-fastest, r[(k-1) + Nz*((i-1) + Nx*(j-1))] (the layout a unit-stride R2C along z
// wants): the divergence is evaluated with threads along x (coalesced reads), transposed through an LDS tile and written with threads
// along z. Same expression as source_term_kernel (weight_by_dz = false: z Periodic).
__global__ void __launch_bounds__(256) source_term_zfast_kernel(DGrid g, FView u, FView v, FView w, double *r) {
    __shared__ double tile[32][33];
    const int i0 = 1 + blockIdx.x * 32, k0 = 1 + blockIdx.y * 32, j = 1 + blockIdx.z;
    const int tx = threadIdx.x, ty = threadIdx.y;          // block (32, 8)
    for (int kk = ty; kk < 32; kk += 8) {
        const int i = i0 + tx, k = k0 + kk;
        if (i <= g.Nx && k <= g.Nz) {
            const int kq = k - 1 + g.Hz;
            const double ax = g.ax[kq], ay = g.ay[kq], az = g.az;
            const double dx = ax * u.at(i + 1, j, k) - ax * u.at(i, j, k);
            const double dy = ay * v.at(i, j + 1, k) - ay * v.at(i, j, k);
            const double dz = az * w.at(i, j, k + 1) - az * w.at(i, j, k);
            const double div = g.vinv_c[kq] * ((dx + dy) + dz);
            tile[kk][tx] = 1.0 * div;
        }
    }
    __syncthreads();
    for (int ii = ty; ii < 32; ii += 8) {
        const int k = k0 + tx, i = i0 + ii;
        if (i <= g.Nx && k <= g.Nz) r[(long)(k - 1) + (long)g.Nz * ((i - 1) + (long)g.Nx * (j - 1))] = tile[tx][ii];
    }
}

// the way back: dense z-fastest real array -> interior of the haloed (Center, Center, Center) field
__global__ void __launch_bounds__(256) copy_real_zfast_kernel(DGrid g, FView p, const double *r) {
    __shared__ double tile[32][33];
    const int i0 = 1 + blockIdx.x * 32, k0 = 1 + blockIdx.y * 32, j = 1 + blockIdx.z;
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int ii = ty; ii < 32; ii += 8) {
        const int k = k0 + tx, i = i0 + ii;
        if (i <= g.Nx && k <= g.Nz) tile[ii][tx] = r[(long)(k - 1) + (long)g.Nz * ((i - 1) + (long)g.Nx * (j - 1))];
    }
    __syncthreads();
    for (int kk = ty; kk < 32; kk += 8) {
        const int i = i0 + tx, k = k0 + kk;
        if (i <= g.Nx && k <= g.Nz) p.at(i, j, k) = tile[tx][kk];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the x-slab pressure step without the passes between its stages (z Periodic, substructured x solve, z-fastest layout).
//   * the source term reads its periodic neighbours at wrapped interior indices and u[Nx+1] from the receive buffer of the one-column
//     exchange: no local fill, no unpack before it;
//   * the pressure correction reads the z-fastest real solution the inverse transform leaves (tiles transposed through LDS, the row
//     j - 1 kept from the previous iteration of a short march along y) and p[0] from the receive buffer: no copy_real pass, no haloed
//     copy of p dt, no fill / unpack of it; it writes u, v, w and p / dt like pressure_correction_kernel (same expressions);
//   * the pack of the early exchange reads wrapped interior rows / levels: no local fill of the strips before it.
// ---------------------------------------------------------------------------------------------------------------------
// one interior column of a haloed field -> dense (Ny, Nz) buffers: west[j-1 + Ny (k-1)] = f[iw, j, k], east[...] = f[ie, j, k]
__global__ void __launch_bounds__(256) column_pack_kernel(DGrid g, FView f, int iw, int ie, double *west, double *east) {
    const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x, k = 1 + blockIdx.y;
    if (j > g.Ny || k > g.Nz) return;
    const long b = (long)(j - 1) + (long)g.Ny * (k - 1);
    west[b] = f.at(iw, j, k);
    east[b] = f.at(ie, j, k);
}
// the same from the z-fastest dense real array r[(k-1) + Nz ((i-1) + Nx (j-1))]: buffers laid out [k-1 + Nz (j-1)]
__global__ void __launch_bounds__(256) column_pack_zfast_kernel(int Nx, int Ny, int Nz, const double *r, double *west, double *east) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (k >= Nz || j >= Ny) return;
    const long b = (long)k + (long)Nz * j;
    west[b] = r[(long)k + (long)Nz * (0 + (long)Nx * j)];
    east[b] = r[(long)k + (long)Nz * ((Nx - 1) + (long)Nx * j)];
}

// source_term_zfast_kernel without halos: y / z neighbours at wrapped interior indices, u[Nx+1, j, k] = ue[j-1 + Ny (k-1)] (the east
// neighbour's first column, received by the one-column exchange). x connected, y and z Periodic.
__global__ void __launch_bounds__(256) source_term_zfast_wrapped_kernel(DGrid g, FView u, FView v, FView w, const double *ue, double *r) {
    __shared__ double tile[32][33];
    const int i0 = 1 + blockIdx.x * 32, k0 = 1 + blockIdx.y * 32, j = 1 + blockIdx.z;
    const int tx = threadIdx.x, ty = threadIdx.y;          // block (32, 8)
    const int jp = j == g.Ny ? 1 : j + 1;
    for (int kk = ty; kk < 32; kk += 8) {
        const int i = i0 + tx, k = k0 + kk;
        if (i <= g.Nx && k <= g.Nz) {
            const int kq = k - 1 + g.Hz, kp = k == g.Nz ? 1 : k + 1;
            const double ax = g.ax[kq], ay = g.ay[kq], az = g.az;
            const double up = i == g.Nx ? ue[(long)(j - 1) + (long)g.Ny * (k - 1)] : u.at(i + 1, j, k);
            const double dx = ax * up - ax * u.at(i, j, k);
            const double dy = ay * v.at(i, jp, k) - ay * v.at(i, j, k);
            const double dz = az * w.at(i, j, kp) - az * w.at(i, j, k);
            const double div = g.vinv_c[kq] * ((dx + dy) + dz);
            tile[kk][tx] = 1.0 * div;
        }
    }
    __syncthreads();
    for (int ii = ty; ii < 32; ii += 8) {
        const int k = k0 + tx, i = i0 + ii;
        if (i <= g.Nx && k <= g.Nz) r[(long)(k - 1) + (long)g.Nz * ((i - 1) + (long)g.Nx * (j - 1))] = tile[tx][ii];
    }
}

// _make_pressure_correction! + `pNHS ./= Δt⁺` (pressure_correction.jl:31-50) from the z-fastest dense solution pd[(k-1) + Nz ((i-1) +
// Nx (j-1))] = p Δt, columns ia .. ib. A block owns a 32 (x) x 32 (z) tile and marches over JB rows of y: the tile of row j (with one
// more column on its low x side and one more level on its low z side) is transposed through LDS, the tile of row j - 1 is the previous
// iteration's. x: p[0, j, k] = pw[k-1 + Nz (j-1)] (the west neighbour's last column, received); y, z Periodic: wrapped.
// Same expressions as pressure_correction_kernel.
#define OCN_ZC_JB 8
__global__ void __launch_bounds__(256) pressure_correction_zfast_kernel(DGrid g, FView u, FView v, FView w, const double *pd, const double *pw,
                                                                        FView p, double divisor, int ia, int ib) {
    __shared__ double T[2][33][33];                   // [buffer][x: i0-1 .. i0+31][z: k0-1 .. k0+31]; odd pitch: no bank conflicts either way
    const int i0 = ia + blockIdx.x * 32, k0 = 1 + blockIdx.y * 32, jfirst = 1 + blockIdx.z * OCN_ZC_JB;
    const int tx = threadIdx.x, ty = threadIdx.y;     // block (32, 8)
    const long sNz = g.Nz, sNx = g.Nx;
    const int ni = min(min(ib, g.Nx), i0 + 31) - (i0 - 1) + 1;      // columns i0 - 1 .. min(ib, Nx, i0 + 31) of the tile
    auto load = [&](int buf, int j) {                 // threads along z (the fast index of pd)
        const int jw = j < 1 ? g.Ny : j;
        for (int ii = ty; ii < ni; ii += 8) {
            const int i = i0 - 1 + ii;
            for (int kk = tx; kk < 33; kk += 32) {
                int k = k0 - 1 + kk;
                if (k > g.Nz) continue;
                if (k < 1) k = g.Nz;
                T[buf][ii][kk] = i < 1 ? pw[(long)(k - 1) + sNz * (jw - 1)] : pd[(long)(k - 1) + sNz * ((i - 1) + sNx * (jw - 1))];
            }
        }
    };
    load(0, jfirst - 1);
    int cur = 1;
    for (int j = jfirst; j < jfirst + OCN_ZC_JB && j <= g.Ny; ++j, cur ^= 1) {
        __syncthreads();                              // the previous iteration's readers of T[cur] are done
        load(cur, j);
        __syncthreads();
        const int i = i0 + tx;
        if (i <= ib && i <= g.Nx) {
            // the three fields may alias as far as the compiler knows: load the four levels of u, v, w first, then store
            double uo[4], vo[4], wo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = min(k0 + ty + 8 * q, g.Nz);
                uo[q] = u.at(i, j, k); vo[q] = v.at(i, j, k); wo[q] = w.at(i, j, k);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kk = ty + 8 * q, k = k0 + kk;
                if (k > g.Nz) continue;
                const double pc = T[cur][tx + 1][kk + 1];
                p.at(i, j, k) = pc / divisor;
                u.at(i, j, k) = uo[q] - (pc - T[cur][tx][kk + 1]) * g.rdx;
                v.at(i, j, k) = vo[q] - (pc - T[cur ^ 1][tx + 1][kk + 1]) * g.rdy;
                w.at(i, j, k) = wo[q] - (pc - T[cur][tx + 1][kk]) * g.rdzf[k - 1 + g.Hz];
            }
        }
    }
}

// the same for the two Hx-wide boundary strips (columns 1 .. H and Nx - H + 1 .. Nx) in one launch: a tile kernel would leave 29 of 32
// lanes idle, so here a thread owns one cell and reads its four pressure values from the z-fastest array directly -- threads along
// (column, y) like pressure_correction_kernel on a strip range; the lines of pd are shared by 16 consecutive levels (L2)
__global__ void __launch_bounds__(256) pressure_correction_zfast_strips_kernel(DGrid g, FView u, FView v, FView w, const double *pd, const double *pw,
                                                                               FView p, double divisor, int H) {
    const int c = threadIdx.x, j = 1 + blockIdx.y * blockDim.y + threadIdx.y, k = 1 + blockIdx.z;
    if (c >= 2 * H || j > g.Ny || k > g.Nz) return;
    const int i = c < H ? 1 + c : g.Nx - 2 * H + 1 + c;
    const long sNz = g.Nz, sNx = g.Nx;
    const int jm = j == 1 ? g.Ny : j - 1, km = k == 1 ? g.Nz : k - 1;
    const double pc = pd[(long)(k - 1) + sNz * ((i - 1) + sNx * (j - 1))];
    const double pim = i == 1 ? pw[(long)(k - 1) + sNz * (j - 1)] : pd[(long)(k - 1) + sNz * ((i - 2) + sNx * (j - 1))];
    const double pjm = pd[(long)(k - 1) + sNz * ((i - 1) + sNx * (jm - 1))];
    const double pkm = pd[(long)(km - 1) + sNz * ((i - 1) + sNx * (j - 1))];
    p.at(i, j, k) = pc / divisor;
    u.at(i, j, k) -= (pc - pim) * g.rdx;
    v.at(i, j, k) -= (pc - pjm) * g.rdy;
    w.at(i, j, k) -= (pc - pkm) * g.rdzf[k - 1 + g.Hz];
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the substructured x solve in the FIELDS' OWN layout (x fastest) -- "xfast". The z-fastest variant above pays for its layout
// twice: the source term and the solution are transposed through LDS (the solution a second time when the pressure correction reads it),
// and the Thomas sweeps along x touch the spectrum twice in each direction. Here nothing is transposed:
//   real rhs / solution  r[(i-1) + Nx ((j-1) + Ny (k-1))]           (what source_term_kernel<real> writes, what the dense correction reads)
//   spectrum             S[i + Nx (ky + Ny kz)],  kz = 0 .. Nz/2      (x lines contiguous: 4 KB at Nx = 256)
// * z: the real-to-complex transform of TWO neighbouring x columns at once -- (r[2i'], r[2i'+1]) read as one complex number, a complex
//   radix-4 line FFT along z in LDS (the machinery of strided_line_fft_kernel: ZL pairs = ZL x 16 B rows per workgroup), and the two
//   columns' half spectra separated with the Hermitian symmetry WHILE THE LINE IS IN LDS (paired_zline_r2c_kernel / _c2r_kernel);
// * y: strided_line_fft_kernel, as on the single-GPU path;
// * x: one WAVE per line. A lane holds E consecutive elements; the two Thomas recurrences f_i = (r_i - a f_{i-1}) d_i and
//   y_i = f_i - a d_i y_{i+1} are affine maps, composed across the 64 lanes by a shuffle scan (6 steps), then re-evaluated inside the lane
//   with the carried-in value -- the whole solve on ONE read of the line. The payload pass (first / last value per mode) reads only; the
//   solve pass after the all-gather runs the same kernel on the right-hand side with the interface values in its two end entries
//   (T p = r - a gL e_0 - a gR e_{N-1}) and writes the final solution: no correction pass, no s = T^-1 e_0 array.
// Same solution as the z-fastest variant to round-off (the scan re-associates the recurrences); validated at creation and by the
// partitioned-vs-single-GPU tests at 1e-12.
// ---------------------------------------------------------------------------------------------------------------------
#define OCN_FFT_CMUL(ar, ai, w) make_double2((ar) * (w).x - (ai) * (w).y, (ar) * (w).y + (ai) * (w).x)
#define OCN_FFT_CMULC(v, w) make_double2((v).x * (w).x + (v).y * (w).y, (v).y * (w).x - (v).x * (w).y)
// forward radix-4 DIF of the ZL lines in zbuf[N][ZL] (natural order in, frequency f at position bitreverse(f) out)
template <int ZL> __device__ __forceinline__ void lds_fft_forward(double2 *zbuf, const double2 *tw, int N, int logn, int il, int kq) {
    const int half = N >> 1, quarter = N >> 2, KQ = 256 / ZL;
#define ZB(n) zbuf[(n) * ZL + il]
    int h = half, st = 1;
    if (logn & 1) {
        for (int q = kq; q < half; q += KQ) {
            const int jj = q & (h - 1), a = ((q - jj) << 1) + jj, b = a + h;
            const double2 xa = ZB(a), xb = ZB(b), w = tw[jj * st];
            ZB(a) = make_double2(xa.x + xb.x, xa.y + xb.y);
            ZB(b) = OCN_FFT_CMUL(xa.x - xb.x, xa.y - xb.y, w);
        }
        __syncthreads();
        h >>= 1; st <<= 1;
    }
    for (; h >= 2; h >>= 2, st <<= 2) {
        const int h2 = h >> 1;
        for (int q = kq; q < quarter; q += KQ) {
            const int jj = q & (h2 - 1), a = ((q - jj) << 2) + jj;
            const double2 x0 = ZB(a), x1 = ZB(a + h2), x2 = ZB(a + h), x3 = ZB(a + h + h2);
            const double2 wa = tw[jj * st], wb = tw[(jj + h2) * st], wc = tw[jj * 2 * st];
            const double2 y0 = make_double2(x0.x + x2.x, x0.y + x2.y), y2 = OCN_FFT_CMUL(x0.x - x2.x, x0.y - x2.y, wa);
            const double2 y1 = make_double2(x1.x + x3.x, x1.y + x3.y), y3 = OCN_FFT_CMUL(x1.x - x3.x, x1.y - x3.y, wb);
            ZB(a) = make_double2(y0.x + y1.x, y0.y + y1.y);
            ZB(a + h2) = OCN_FFT_CMUL(y0.x - y1.x, y0.y - y1.y, wc);
            ZB(a + h) = make_double2(y2.x + y3.x, y2.y + y3.y);
            ZB(a + h + h2) = OCN_FFT_CMUL(y2.x - y3.x, y2.y - y3.y, wc);
        }
        __syncthreads();
    }
#undef ZB
}
// inverse radix-4 DIT (frequency f at position bitreverse(f) in, natural order out, unnormalised)
template <int ZL> __device__ __forceinline__ void lds_fft_inverse(double2 *zbuf, const double2 *tw, int N, int il, int kq) {
    const int half = N >> 1, quarter = N >> 2, KQ = 256 / ZL;
#define ZB(n) zbuf[(n) * ZL + il]
    int h = 1, st = half;
    for (; (h << 1) <= half; h <<= 2, st >>= 2) {
        for (int q = kq; q < quarter; q += KQ) {
            const int jj = q & (h - 1), a = ((q - jj) << 2) + jj;
            const double2 x0 = ZB(a), x1 = ZB(a + h), x2 = ZB(a + 2 * h), x3 = ZB(a + 3 * h);
            const double2 wa = tw[jj * st], wb = tw[jj * (st >> 1)], wc = tw[(jj + h) * (st >> 1)];
            const double2 t1 = OCN_FFT_CMULC(x1, wa), t3 = OCN_FFT_CMULC(x3, wa);
            const double2 y0 = make_double2(x0.x + t1.x, x0.y + t1.y), y1 = make_double2(x0.x - t1.x, x0.y - t1.y);
            const double2 y2 = make_double2(x2.x + t3.x, x2.y + t3.y), y3 = make_double2(x2.x - t3.x, x2.y - t3.y);
            const double2 u2 = OCN_FFT_CMULC(y2, wb), u3 = OCN_FFT_CMULC(y3, wc);
            ZB(a) = make_double2(y0.x + u2.x, y0.y + u2.y);
            ZB(a + 2 * h) = make_double2(y0.x - u2.x, y0.y - u2.y);
            ZB(a + h) = make_double2(y1.x + u3.x, y1.y + u3.y);
            ZB(a + 3 * h) = make_double2(y1.x - u3.x, y1.y - u3.y);
        }
        __syncthreads();
    }
    if (h <= half) {
        for (int q = kq; q < half; q += KQ) {
            const int jj = q & (h - 1), a = ((q - jj) << 1) + jj, b = a + h;
            const double2 xa = ZB(a), t = OCN_FFT_CMULC(ZB(b), tw[jj * st]);
            ZB(a) = make_double2(xa.x + t.x, xa.y + t.y);
            ZB(b) = make_double2(xa.x - t.x, xa.y - t.y);
        }
        __syncthreads();
    }
#undef ZB
}

// real-to-complex along z of the column PAIRS of a dense x-fastest real array: rp[c + C k] = (r[2c], r[2c+1]) at level k, c = pair index
// over (x, y) (C = Nx Ny / 2 pairs per level); out: spec[2c + 2C kz], spec[2c + 1 + 2C kz], kz = 0 .. N/2 -- the half spectra of the two
// columns. With Z = FFT(r_even + i r_odd): X_even[k] = (Z[k] + conj Z[N-k]) / 2, X_odd[k] = (Z[k] - conj Z[N-k]) / (2i).
template <int ZL>
__global__ void __launch_bounds__(256) paired_zline_r2c_kernel(const double2 *rp, double2 *spec, const double2 *tw, long C, int N, int logn) {
    extern __shared__ double2 zbuf[];                 // [N][ZL]
    const int il = threadIdx.x % ZL, kq = threadIdx.x / ZL, KQ = 256 / ZL;
    const long c = (long)blockIdx.x * ZL + il;
    const bool live = c < C;
    for (int k = kq; k < N; k += KQ) zbuf[k * ZL + il] = live ? rp[c + C * k] : make_double2(0.0, 0.0);
    __syncthreads();
    lds_fft_forward<ZL>(zbuf, tw, N, logn, il, kq);
    if (!live) return;
    for (int kz = kq; kz <= (N >> 1); kz += KQ) {
        const int pa = (int)(__brev((unsigned)kz) >> (32 - logn)), pb = (int)(__brev((unsigned)((N - kz) & (N - 1))) >> (32 - logn));
        const double2 a = zbuf[pa * ZL + il], b = zbuf[pb * ZL + il];
        double2 *o = spec + 2 * c + 2 * C * (long)kz;
        o[0] = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y - b.y));
        o[1] = make_double2(0.5 * (a.y + b.y), -0.5 * (a.x - b.x));
    }
}
// the way back: Z[k] = X_even[k] + i X_odd[k], Z[N-k] = conj X_even[k] + i conj X_odd[k]; the imaginary parts of the k = 0 and k = N/2
// entries are ignored like a complex-to-real transform ignores them; rp = Re / Im of the inverse transform, times `scale`
template <int ZL>
__global__ void __launch_bounds__(256) paired_zline_c2r_kernel(const double2 *spec, double2 *rp, const double2 *tw, long C, int N, int logn, double scale) {
    extern __shared__ double2 zbuf[];
    const int il = threadIdx.x % ZL, kq = threadIdx.x / ZL, KQ = 256 / ZL;
    const long c = (long)blockIdx.x * ZL + il;
    const bool live = c < C;
    for (int kz = kq; kz <= (N >> 1); kz += KQ) {
        const double2 *in = spec + 2 * c + 2 * C * (long)kz;
        const double2 E = live ? in[0] : make_double2(0.0, 0.0), O = live ? in[1] : make_double2(0.0, 0.0);
        const int pa = (int)(__brev((unsigned)kz) >> (32 - logn));
        if (kz == 0 || kz == (N >> 1)) zbuf[pa * ZL + il] = make_double2(E.x, O.x);
        else {
            const int pb = (int)(__brev((unsigned)(N - kz)) >> (32 - logn));
            zbuf[pa * ZL + il] = make_double2(E.x - O.y, E.y + O.x);
            zbuf[pb * ZL + il] = make_double2(E.x + O.y, O.x - E.y);
        }
    }
    __syncthreads();
    lds_fft_inverse<ZL>(zbuf, tw, N, il, kq);
    if (!live) return;
    for (int k = kq; k < N; k += KQ) {
        const double2 v = zbuf[k * ZL + il];
        rp[c + C * k] = make_double2(v.x * scale, v.y * scale);
    }
}

// Thomas factors of every mode in the x-fastest layout: rden[i + N m] = 1 / (b_m - a cp[i-1]), cp = a rden (the recurrences of
// sub_setup_kernel), and of s = T^-1 e_0 only what the interface system needs: its first and last entry (and its sum for the null mode).
// `scratch` (N M doubles) holds s while it is back-substituted. One thread per mode; runs once.
__global__ void __launch_bounds__(256) sub_setup_xfast_kernel(long M, int n0, int N, double a, const double *l0, const double *l1, double *rden,
                                                              double *s_first, double *s_last, double *ssum0, double *scratch) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const double b = -2.0 * a - (l0[m % n0] + l1[m / n0]);
    double *rd = rden + (long)N * m, *sv = scratch + (long)N * m;
    double c = 0.0;
    for (int i = 0; i < N; ++i) {
        const double r = 1.0 / (b - a * c);
        c = a * r;
        rd[i] = r;
    }
    double y = rd[0];
    sv[0] = y;
    for (int i = 1; i < N; ++i) { y = (-a * y) * rd[i]; sv[i] = y; }
    for (int i = N - 2; i >= 0; --i) sv[i] -= (a * rd[i]) * sv[i + 1];
    s_first[m] = sv[0];
    s_last[m] = sv[N - 1];
    if (m == 0) {
        double t = 0.0;
        for (int i = 0; i < N; ++i) t += sv[i];
        *ssum0 = t;
    }
}

// W lanes per x line (mode m; W = 64: one wave per line, W = 8 / 16 / 32: the 8 / 4 / 2 lines of a wave for the short lines of thin slabs --
// with one element per lane a 64-point line spent its time in the six scan steps: 0.115 ms per pass on a 64 x 512 x 512 slab): the Dirichlet
// solve y = T^-1 r by Thomas, both recurrences as shuffle scans of affine maps over the W lanes (see the section comment).
// SOLVE = false: reads only, payload[m] = y[0], payload[M + m] = y[N-1], payload[2M] = sum_i y[i] of mode 0.
// SOLVE = true: r_0 -= a gL, r_{N-1} -= a gR (iface[m], iface[M + m]) first, writes (y - mean[mode 0]) * scale in place.
template <int E, bool SOLVE, int W = 64>
__global__ void __launch_bounds__(256) xline_thomas_kernel(double2 *S, const double *__restrict__ rden, long M, int N, double a, double2 *payload,
                                                           const double2 *__restrict__ iface, double scale) {
    constexpr int LPW = 64 / W;                        // lines per wave
    const int lane = threadIdx.x & 63, sl = lane & (W - 1);
    const long mw = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * LPW;
    if (mw >= M) return;                               // whole waves leave together
    const long mr = mw + lane / W;
    const bool live = mr < M;                          // lines beyond the last one: lanes repeat the last line, write nothing
    const long m = live ? mr : M - 1;
    const int i0 = sl * E;
    double2 *line = S + (long)N * m;
    const double *dl = rden + (long)N * m;
    double2 r[E], f[E];
    double d[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool in = i0 + e < N;
        r[e] = in ? line[i0 + e] : make_double2(0.0, 0.0);
        d[e] = in ? dl[i0 + e] : 0.0;                  // beyond the line: both maps are the zero map
    }
    if (SOLVE) {
        const double2 gl = iface[m], gr = iface[M + m];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (i0 + e == 0) { r[e].x -= a * gl.x; r[e].y -= a * gl.y; }
            if (i0 + e == N - 1) { r[e].x -= a * gr.x; r[e].y -= a * gr.y; }
        }
    }
    // forward: f_i = A_i f_{i-1} + B_i, A_i = -a d_i, B_i = r_i d_i
    double A = 1.0, Bx = 0.0, By = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double Ae = -a * d[e];
        Bx = Ae * Bx + r[e].x * d[e]; By = Ae * By + r[e].y * d[e];
        A = Ae * A;
    }
#pragma unroll
    for (int off = 1; off < W; off <<= 1) {
        const double Ap = __shfl_up(A, off, W), Bpx = __shfl_up(Bx, off, W), Bpy = __shfl_up(By, off, W);
        if (sl >= off) { Bx = A * Bpx + Bx; By = A * Bpy + By; A = A * Ap; }
    }
    double2 prev = make_double2(__shfl_up(Bx, 1, W), __shfl_up(By, 1, W));
    if (sl == 0) prev = make_double2(0.0, 0.0);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        f[e] = make_double2((r[e].x - a * prev.x) * d[e], (r[e].y - a * prev.y) * d[e]);
        prev = f[e];
    }
    // backward: y_i = f_i - (a d_i) y_{i+1}
    A = 1.0; Bx = 0.0; By = 0.0;
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
        const double Ae = -(a * d[e]);
        Bx = Ae * Bx + f[e].x; By = Ae * By + f[e].y;
        A = Ae * A;
    }
#pragma unroll
    for (int off = 1; off < W; off <<= 1) {
        const double Ap = __shfl_down(A, off, W), Bpx = __shfl_down(Bx, off, W), Bpy = __shfl_down(By, off, W);
        if (sl + off < W) { Bx = A * Bpx + Bx; By = A * Bpy + By; A = A * Ap; }
    }
    prev = make_double2(__shfl_down(Bx, 1, W), __shfl_down(By, 1, W));
    if (sl == W - 1) prev = make_double2(0.0, 0.0);
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
        const double c = a * d[e];
        f[e] = make_double2(f[e].x - c * prev.x, f[e].y - c * prev.y);       // f now holds y
        prev = f[e];
    }
    if (SOLVE) {
        const double2 mu = m == 0 ? iface[2 * M] : make_double2(0.0, 0.0);
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (live && i0 + e < N) line[i0 + e] = make_double2((f[e].x - mu.x) * scale, (f[e].y - mu.y) * scale);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (live && i0 + e == 0) payload[m] = f[e];
            if (live && i0 + e == N - 1) payload[M + m] = f[e];
        }
        if (mw == 0) {                                 // the wave that holds mode 0 (its first W lanes): the sum over that line
            double sx = 0.0, sy = 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) { sx += f[e].x; sy += f[e].y; }        // entries beyond the line are zero
            for (int off = W / 2; off > 0; off >>= 1) { sx += __shfl_down(sx, off, W); sy += __shfl_down(sy, off, W); }
            if (lane == 0) payload[2 * M] = make_double2(sx, sy);
        }
    }
}

// known-answer check of the paired z transform at solver creation: pair c, level k holds (cos(2 pi k / N), sin(2 pi 3 k / N)) scaled by
// a pair-dependent amplitude; expected: X_even[1] = amp N/2, X_odd[3] = -i amp N/2, zero elsewhere (N >= 8; sampled on 256 pairs)
__device__ __forceinline__ double xfast_kat_amp(long c) { return 1.0 + 0.25 * (double)(c % 7); }
__global__ void __launch_bounds__(256) xfast_kat_fill_kernel(double2 *rp, long C, int N) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C * N) return;
    const long c = t % C;
    const int k = (int)(t / C);
    double s3, c3, s5, c5;
    sincospi(2.0 * 1.0 * (double)k / (double)N, &s3, &c3);
    sincospi(2.0 * 3.0 * (double)k / (double)N, &s5, &c5);
    (void)s3; (void)c5;
    rp[t] = make_double2(xfast_kat_amp(c) * c3, xfast_kat_amp(c) * s5);
}
__global__ void __launch_bounds__(256) xfast_kat_check_kernel(const double2 *spec, long C, int N, double *out) {
    __shared__ double sm[256];
    const long c = (C - 1) * (long)threadIdx.x / 255;               // 256 pairs spread over the array
    double worst = 0.0;
    for (int kz = 0; kz <= N / 2; ++kz) {
        const double2 E = spec[2 * c + 2 * C * (long)kz], O = spec[2 * c + 1 + 2 * C * (long)kz];
        const double amp = xfast_kat_amp(c);
        const double2 eE = kz == 1 ? make_double2(amp * N / 2.0, 0.0) : make_double2(0.0, 0.0);
        const double2 eO = kz == 3 ? make_double2(0.0, -amp * N / 2.0) : make_double2(0.0, 0.0);
        worst = fmax(worst, fmax(fmax(fabs(E.x - eE.x), fabs(E.y - eE.y)), fmax(fabs(O.x - eO.x), fabs(O.y - eO.y))));
    }
    sm[threadIdx.x] = worst;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sm[0];
}
__global__ void __launch_bounds__(256) xfast_kat_check_real_kernel(const double2 *rp, long C, int N, double *out) {
    __shared__ double sm[256];
    const long c = (C - 1) * (long)threadIdx.x / 255;
    double worst = 0.0;
    for (int k = 0; k < N; ++k) {
        double s3, c3, s5, c5;
        sincospi(2.0 * 1.0 * (double)k / (double)N, &s3, &c3);
        sincospi(2.0 * 3.0 * (double)k / (double)N, &s5, &c5);
        (void)s3; (void)c5;
        const double2 v = rp[c + C * (long)k];
        worst = fmax(worst, fmax(fabs(v.x - xfast_kat_amp(c) * c3), fabs(v.y - xfast_kat_amp(c) * s5)));
    }
    sm[threadIdx.x] = worst;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sm[0];
}
// dense x-fastest real array -> interior of the haloed (Center, Center, Center) field
__global__ void __launch_bounds__(256) copy_dense_to_field_kernel(DGrid g, FView phi, const double *r) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    phi.at(i, j, k) = r[(long)(i - 1) + (long)g.Nx * ((j - 1) + (long)g.Ny * (k - 1))];
}

// compute_source_term! into the dense x-fastest real array with NO halo read: y / z neighbours at wrapped interior indices, u[Nx+1, j, k]
// from `ue` (the east neighbour's first column, dense (Ny, Nz)); expression of source_value (weight_by_dz = false)
__global__ void __launch_bounds__(256) source_term_dense_wrapped_kernel(DGrid g, FView u, FView v, FView w, const double *ue, double *r) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    const int kq = k - 1 + g.Hz, jp = j == g.Ny ? 1 : j + 1, kp = k == g.Nz ? 1 : k + 1;
    const double ax = g.ax[kq], ay = g.ay[kq], az = g.az;
    const double up = i == g.Nx ? ue[(long)(j - 1) + (long)g.Ny * (k - 1)] : u.at(i + 1, j, k);
    const double dx = ax * up - ax * u.at(i, j, k);
    const double dy = ay * v.at(i, jp, k) - ay * v.at(i, j, k);
    const double dz = az * w.at(i, j, kp) - az * w.at(i, j, k);
    const double div = g.vinv_c[kq] * ((dx + dy) + dz);
    r[(long)(i - 1) + (long)g.Nx * ((j - 1) + (long)g.Ny * (k - 1))] = 1.0 * div;
}

// source_term_dense_wrapped_kernel + paired_zline_r2c_kernel in ONE pass: the divergences of a column pair go straight into the LDS line
// buffer of the z transform instead of through the dense real array (a write and a read of N^3 doubles less per solve: 0.105 + 0.075 -> 0.153 ms
// at 256^3; unrolling the level loop changes nothing). Same expressions on
// the same operands as the two kernels => the same bits. Pair c covers the columns (2c, 2c + 1) of the (x, y) plane, x fastest (Nx even).
template <int ZL>
__global__ void __launch_bounds__(256) source_paired_zline_r2c_kernel(DGrid g, FView u, FView v, FView w, const double *ue, double2 *spec, const double2 *tw,
                                                                      long C, int N, int logn) {
    extern __shared__ double2 zbuf[];                 // [N][ZL]
    const int il = threadIdx.x % ZL, kq = threadIdx.x / ZL, KQ = 256 / ZL;
    const long c = (long)blockIdx.x * ZL + il;
    const bool live = c < C;
    const int hx = g.Nx >> 1;
    const int j = live ? (int)(c / hx) + 1 : 1, i0 = live ? 2 * (int)(c % hx) + 1 : 1;
    const int jp = j == g.Ny ? 1 : j + 1;
    const double az = g.az;
    for (int kk = kq; kk < N; kk += KQ) {
        const int k = kk + 1, kt = k - 1 + g.Hz, kp = k == g.Nz ? 1 : k + 1;
        const double ax = g.ax[kt], ay = g.ay[kt], vinv = g.vinv_c[kt];
        const double u0 = u.at(i0, j, k), u1 = u.at(i0 + 1, j, k);
        const double u2 = (i0 + 1 == g.Nx && ue) ? ue[(long)(j - 1) + (long)g.Ny * (k - 1)] : u.at(i0 + 2, j, k);
        const double v0 = v.at(i0, j, k), v1 = v.at(i0 + 1, j, k), v0p = v.at(i0, jp, k), v1p = v.at(i0 + 1, jp, k);
        const double w0 = w.at(i0, j, k), w1 = w.at(i0 + 1, j, k), w0p = w.at(i0, j, kp), w1p = w.at(i0 + 1, j, kp);
        const double d0 = vinv * (((ax * u1 - ax * u0) + (ay * v0p - ay * v0)) + (az * w0p - az * w0));
        const double d1 = vinv * (((ax * u2 - ax * u1) + (ay * v1p - ay * v1)) + (az * w1p - az * w1));
        zbuf[kk * ZL + il] = live ? make_double2(1.0 * d0, 1.0 * d1) : make_double2(0.0, 0.0);
    }
    __syncthreads();
    lds_fft_forward<ZL>(zbuf, tw, N, logn, il, kq);
    if (!live) return;
    for (int kz = kq; kz <= (N >> 1); kz += KQ) {
        const int pa = (int)(__brev((unsigned)kz) >> (32 - logn)), pb = (int)(__brev((unsigned)((N - kz) & (N - 1))) >> (32 - logn));
        const double2 a = zbuf[pa * ZL + il], b = zbuf[pb * ZL + il];
        double2 *o = spec + 2 * c + 2 * C * (long)kz;
        o[0] = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y - b.y));
        o[1] = make_double2(0.5 * (a.y + b.y), -0.5 * (a.x - b.x));
    }
}

// first / last column of the dense x-fastest array -> dense (Ny, Nz) buffers [j-1 + Ny (k-1)]
__global__ void __launch_bounds__(256) column_pack_dense_kernel(int Nx, int Ny, int Nz, const double *r, double *west, double *east, long sj, long sk) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (j >= Ny || k >= Nz) return;
    const long row = sj * j + sk * k, b = (long)j + (long)Ny * k;
    west[b] = r[row];
    east[b] = r[row + Nx - 1];
}

// pressure_correction_dense_kernel on the columns ia .. ib of a partitioned slab: p[0, j, k] = pw[j-1 + Ny (k-1)] (the west neighbour's
// last column, received) instead of the periodic wrap; y and z Periodic. Same expressions.
// The dense array holds cell (i, j, k) at (i-1) + sj (j-1) + sk (k-1): (Nx, Nx Ny) for the x-fastest substructured solve, (Nxe Nz, Nxe) for the
// transposing solvers' paired-column layout; zbounded: p[0] = p[1] (the no-flux halo of a Bounded z: the bottom face keeps its w).
__global__ void __launch_bounds__(256) pressure_correction_dense_slab_kernel(DGrid g, FView u, FView v, FView w, const double *pd, const double *pw,
                                                                             FView p, double divisor, int ia, int ib, long sj, long sk, bool zbounded,
                                                                             bool store_p) {
    const int i = ia + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > ib || j > g.Ny || k > g.Nz) return;
    const long sx = sj, sxy = sk;
    const long q = (long)(i - 1) + sx * (j - 1) + sxy * (k - 1);
    const double pc = pd[q];
    const double pim = i == 1 ? pw[(long)(j - 1) + (long)g.Ny * (k - 1)] : pd[q - 1];
    const double pjm = pd[j == 1 ? q + sx * (g.Ny - 1) : q - sx];
    const double pkm = k == 1 ? (zbounded ? pc : pd[q + sxy * (g.Nz - 1)]) : pd[q - sxy];
    u.at(i, j, k) -= (pc - pim) * g.rdx;
    v.at(i, j, k) -= (pc - pjm) * g.rdy;
    w.at(i, j, k) -= (pc - pkm) * g.rdzf[k - 1 + g.Hz];
    if (store_p) p.at(i, j, k) = pc / divisor;
}
// ... and on the two Hx-wide boundary strips in one launch (threads along (column, y))
__global__ void __launch_bounds__(256) pressure_correction_dense_strips_kernel(DGrid g, FView u, FView v, FView w, const double *pd, const double *pw,
                                                                               FView p, double divisor, int H, long sj, long sk, bool zbounded, bool store_p) {
    const int c = threadIdx.x, j = 1 + blockIdx.y * blockDim.y + threadIdx.y, k = 1 + blockIdx.z;
    if (c >= 2 * H || j > g.Ny || k > g.Nz) return;
    const int i = c < H ? 1 + c : g.Nx - 2 * H + 1 + c;
    const long sx = sj, sxy = sk;
    const long q = (long)(i - 1) + sx * (j - 1) + sxy * (k - 1);
    const double pc = pd[q];
    const double pim = i == 1 ? pw[(long)(j - 1) + (long)g.Ny * (k - 1)] : pd[q - 1];
    const double pjm = pd[j == 1 ? q + sx * (g.Ny - 1) : q - sx];
    const double pkm = k == 1 ? (zbounded ? pc : pd[q + sxy * (g.Nz - 1)]) : pd[q - sxy];
    u.at(i, j, k) -= (pc - pim) * g.rdx;
    v.at(i, j, k) -= (pc - pjm) * g.rdy;
    w.at(i, j, k) -= (pc - pkm) * g.rdzf[k - 1 + g.Hz];
    if (store_p) p.at(i, j, k) = pc / divisor;
}

// per-block max |a - b| (plan cross-checks)
__global__ void __launch_bounds__(256) max_abs_diff_kernel(const double *a, const double *b, long n, double *blockmax) {
    __shared__ double sm[256];
    double m = 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) m = fmax(m, fabs(a[q] - b[q]));
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) blockmax[blockIdx.x] = sm[0];
}

// the same for two row-major complex arrays of `n0` columns and `rows` rows with different row pitches
__global__ void __launch_bounds__(256) max_abs_diff_pitched_kernel(const double2 *a, int pa, const double2 *b, int pb, int n0, long rows,
                                                                   double *blockmax) {
    __shared__ double sm[256];
    double m = 0;
    const long total = (long)n0 * rows;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long r = q / n0;
        const int c = q - r * n0;
        const double2 x = a[c + (long)pa * r], y = b[c + (long)pb * r];
        m = fmax(m, fmax(fabs(x.x - y.x), fabs(x.y - y.y)));
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) blockmax[blockIdx.x] = sm[0];
}

// deterministic two-stage sum of a complex array (for mean(ϕ)); stage 1: per-block partials, stage 2: one block.
__global__ void __launch_bounds__(256) sum_partial_kernel(const double2 *x, long n, double2 *partial) {
    __shared__ double sx[256], sy[256];
    double ax = 0, ay = 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        ax += x[q].x; ay += x[q].y;
    }
    sx[threadIdx.x] = ax; sy[threadIdx.x] = ay;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sx[threadIdx.x] += sx[threadIdx.x + s]; sy[threadIdx.x] += sy[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = make_double2(sx[0], sy[0]);
}
__global__ void __launch_bounds__(256) sum_final_kernel(const double2 *partial, int nb, double inv_count, double scale, double2 *mean) {
    __shared__ double sx[256], sy[256];
    double ax = 0, ay = 0;
    for (int q = threadIdx.x; q < nb; q += 256) { ax += partial[q].x; ay += partial[q].y; }
    sx[threadIdx.x] = ax; sy[threadIdx.x] = ay;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sx[threadIdx.x] += sx[threadIdx.x + s]; sy[threadIdx.x] += sy[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *mean = make_double2(sx[0] * scale * inv_count, sy[0] * scale * inv_count);
}

// max |∇·u| (test helper)
__global__ void __launch_bounds__(256) max_abs_div_kernel(DGrid g, FView u, FView v, FView w, double *blockmax) {
    __shared__ double sm[256];
    double m = 0;
    long total = (long)g.Nx * g.Ny * g.Nz;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        int i = 1 + q % g.Nx, j = 1 + (q / g.Nx) % g.Ny, k = 1 + q / ((long)g.Nx * g.Ny);
        const int kk = k - 1 + g.Hz;
        double dx = g.tx == OCN_FLAT ? 0.0 : g.ax[kk] * u.at(i + 1, j, k) - g.ax[kk] * u.at(i, j, k);
        double dy = g.ty == OCN_FLAT ? 0.0 : g.ay[kk] * v.at(i, j + 1, k) - g.ay[kk] * v.at(i, j, k);
        double dz = g.tz == OCN_FLAT ? 0.0 : g.az * w.at(i, j, k + 1) - g.az * w.at(i, j, k);
        double d = fabs(g.vinv_c[kk] * ((dx + dy) + dz));
        m = d > m ? d : m;
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) blockmax[blockIdx.x] = sm[0];
}

// cell_advection_timescale (Advection/cell_advection_timescale.jl:13-34): per-block maximum of the inverse timescale
// |u| Δx⁻¹ + |v| Δy⁻¹ + |w| Δzᶠ⁻¹ (min of 1/x = 1 / max of x: the division is monotone, the result has the reference's bits)
__global__ void __launch_bounds__(256) advection_timescale_kernel(DGrid g, FView u, FView v, FView w, double *blockmax) {
    __shared__ double sm[256];
    double m = 0;
    const long total = (long)g.Nx * g.Ny * g.Nz;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int i = 1 + q % g.Nx, j = 1 + (q / g.Nx) % g.Ny, k = 1 + q / ((long)g.Nx * g.Ny);
        const double ix = g.tx == OCN_FLAT ? 0.0 : fabs(u.at(i, j, k)) * (1.0 / g.dx);
        const double iy = g.ty == OCN_FLAT ? 0.0 : fabs(v.at(i, j, k)) * (1.0 / g.dy);
        const double iz = g.tz == OCN_FLAT ? 0.0 : fabs(w.at(i, j, k)) * (1.0 / g.dzf[k - 1 + g.Hz]);
        const double d = (ix + iy) + iz;
        m = d > m ? d : m;
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) blockmax[blockIdx.x] = sm[0];
}

// hasnan(field) = any(isnan, parent(field)) (Diagnostics/nan_checker.jl:32)
__global__ void __launch_bounds__(256) hasnan_kernel(const double *a, long n, int *flag) {
    bool found = false;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) found |= a[q] != a[q];
    if (__any(found) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// exhaustive check of rcp_rn_f32<VARIANT> against the compiler's correctly rounded divide over one binade
template <int VARIANT>
__global__ void __launch_bounds__(256) rcp_check_kernel(int exponent_bits, unsigned long long *mismatches) {
    unsigned m = blockIdx.x * blockDim.x + threadIdx.x;          // 2^23 significands
    if (m >= (1u << 23)) return;
    float x = __uint_as_float(((unsigned)exponent_bits << 23) | m);
    float a = rcp_rn_f32<VARIANT>(x), b = 1.0f / x;
    if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(mismatches, 1ULL);
}

// sampled check of rcp_rn_f64 against the compiler's IEEE divide: pseudo-random significands, exponents exp_lo..exp_hi
__global__ void __launch_bounds__(256) rcp64_check_kernel(unsigned long long n, int exp_lo, int exp_hi, unsigned long long seed,
                                                          unsigned long long *mismatches) {
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    unsigned long long h = (t + seed) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
    const unsigned long long mant = h & ((1ull << 52) - 1);
    const int e = exp_lo + (int)((h >> 52) % (unsigned long long)(exp_hi - exp_lo + 1));
    const double x = __longlong_as_double((long long)(((unsigned long long)(e + 1023) << 52) | mant));
    const double a = rcp_rn_f64(x), b = 1.0 / x;
    if (__double_as_longlong(a) != __double_as_longlong(b)) atomicAdd(mismatches, 1ULL);
}

// ---------------------------------------------------------------------------------------------------------------------
// distributed x-slab support (src/DistributedComputations)
// ---------------------------------------------------------------------------------------------------------------------
// fill_send_buffers! / recv_from_buffers! for a 1-D x partition (communication_buffers.jl:281-313): the west / east
// buffers hold Hx x Py x Pz slabs -- the whole parent extent in y and z, so corners ride along (:53,71-76).
// PACK: west_send <- parent[Hx .. 2Hx), east_send <- parent[Nx .. Nx+Hx);  UNPACK: parent[0 .. Hx) <- west_recv,
// parent[Nx+Hx .. Nx+2Hx) <- east_recv. Buffers are field-major; field f (its own Py x Pz: Face fields on Bounded
// dimensions have one more plane) starts at off[f] and holds rows[f] = Py*Pz rows of Hx values.
struct SlabList {
    long off[OCN_MAX_FIELDS];
    long rows[OCN_MAX_FIELDS];
    int p0[OCN_MAX_FIELDS];                    // parent extent in x of each field
};
template <bool PACK>
__global__ void __launch_bounds__(256) x_halo_buffer_kernel(FieldList fl, SlabList sl, int N, int HX, int H, double *west, double *east,
                                                            bool do_west, bool do_east) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int h = t % H;
    const long r = t / H;                      // j + P1 * k
    for (int f = 0; f < fl.n; ++f) {
        if (r >= sl.rows[f]) continue;
        double *p = fl.p[f];
        const long row = r * sl.p0[f];
        const long b = sl.off[f] + t;
        // H = exchanged depth (<= Hx = HX): the depth interior columns next to each side <-> the depth halo columns nearest to it
        if (PACK) {
            west[b] = p[row + HX + h];
            east[b] = p[row + HX + N - H + h];
        } else {
            if (do_west) p[row + HX - H + h] = west[b];
            if (do_east) p[row + N + HX + h] = east[b];
        }
    }
}

// fill_send_buffers! of x_halo_buffer_kernel<true> with the y / z halo rows read at their WRAPPED interior source (y, z Periodic, every
// field with parent extents (Px, Ny + 2Hy, Nz + 2Hz)): what the buffers would hold after a local fill of the packed columns
__global__ void __launch_bounds__(256) x_halo_pack_wrapped_kernel(FieldList fl, SlabList sl, int N, int HX, int H, int Ny, int Hy, int Nz, int Hz,
                                                                  double *west, double *east) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int h = t % H;
    const long r = t / H;                      // j + P1 * k over the parent extents
    const int P1 = Ny + 2 * Hy;
    const int jp = (int)(r % P1), kp = (int)(r / P1);
    if (kp >= Nz + 2 * Hz) return;
    const int js = Hy + ((jp - Hy) % Ny + Ny) % Ny, ks = Hz + ((kp - Hz) % Nz + Nz) % Nz;
    for (int f = 0; f < fl.n; ++f) {
        const double *p = fl.p[f];
        const long row = ((long)js + (long)P1 * ks) * sl.p0[f];
        const long b = sl.off[f] + t;
        west[b] = p[row + HX + h];
        east[b] = p[row + HX + N - H + h];
    }
}

// the same along y for pencil partitions: the south / north buffers hold Hy rows of the WHOLE parent extent in x and z, so the x halos
// just received -- the corners of halo_communication.jl:137-162 -- ride along on the second hop
struct RowList {
    long off[OCN_MAX_FIELDS];
    int p0[OCN_MAX_FIELDS], p1[OCN_MAX_FIELDS], p2[OCN_MAX_FIELDS];
};
template <bool PACK>
__global__ void __launch_bounds__(256) y_halo_buffer_kernel(FieldList fl, RowList rl, int N, int HY, int H, double *south, double *north,
                                                            bool do_south, bool do_north) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int f = 0; f < fl.n; ++f) {
        const int p0 = rl.p0[f];
        const long per = (long)p0 * H;
        if (t >= per * rl.p2[f]) continue;
        const int i = t % p0, h = (t / p0) % H;
        const long k = t / per;
        double *p = fl.p[f];
        const long base = i + (long)p0 * (long)rl.p1[f] * k;
        const long b = rl.off[f] + t;
        if (PACK) {
            south[b] = p[base + (long)p0 * (HY + h)];
            north[b] = p[base + (long)p0 * (HY + N - H + h)];
        } else {
            if (do_south) p[base + (long)p0 * (HY - H + h)] = south[b];
            if (do_north) p[base + (long)p0 * (N + HY + h)] = north[b];
        }
    }
}

// transpose staging for the distributed FFT (distributed_transpose.jl:25-95), x-slab partition over R ranks:
//   y->x pack : zfield (Nxl, Ny, Nz)   -> send chunk d = { (i, jl, k) : j = d*Nyl + jl },  chunk-major
//   y->x unpack: recv chunk s          -> xfield (Nxg, Nyl, Nz) at i_g = s*Nxl + i
// and the inverse pair. dir 0: zfield -> chunks; 1: chunks -> xfield; 2: xfield -> chunks; 3: chunks -> zfield
__global__ void __launch_bounds__(256) transpose_stage_kernel(int dir, int R, int Nxl, int Nyl, int Nz, const double2 *src,
                                                              double2 *dst) {
    const long chunk = (long)Nxl * Nyl * Nz;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= chunk * R) return;
    const int c = t / chunk;                   // peer rank index
    long q = t % chunk;
    const int i = q % Nxl;
    const int jl = (q / Nxl) % Nyl;
    const int k = q / ((long)Nxl * Nyl);
    const long Ny = (long)Nyl * R, Nxg = (long)Nxl * R;
    const long zidx = i + Nxl * ((long)(c * Nyl + jl) + Ny * k);        // zfield element (i, j = c*Nyl + jl, k)
    const long xidx = ((long)c * Nxl + i) + Nxg * (jl + (long)Nyl * k);  // xfield element (i_g = c*Nxl + i, jl, k)
    if (dir == 0) dst[t] = src[zidx];
    else if (dir == 1) dst[xidx] = src[t];
    else if (dir == 2) dst[t] = src[xidx];
    else dst[zidx] = src[t];
}

// _solve_poisson_in_spectral_space! (distributed_fft_based_poisson_solver.jl:180-188) on the x-local layout
// (Nxg, Nyc, Nz): ϕ̂ = -b̂ / (λx + λy + λz), λy indexed with the rank's mode offset; zeroth mode zeroed on the rank owning
// it. `scale` folds the inverse-transform normalisation. Padding modes (j >= Ny) carry zeros and are clamped.
__global__ void __launch_bounds__(256) dist_spectral_divide_kernel(double2 *b, const double *lx, const double *ly, const double *lz,
                                                                   int Nxg, int Nyc, int Nz, int joff, int Ny, double scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = blockIdx.z;
    if (i >= Nxg || j >= Nyc || k >= Nz) return;
    const long q = (long)i + (long)Nxg * (j + (long)Nyc * k);
    const int jg = min(joff + j, Ny - 1);
    double lam = (lx[i] + ly[jg]) + lz[k] - 0.0;
    double2 val = b[q];
    val.x = -(val.x * scale) / lam;
    val.y = -(val.y * scale) / lam;
    if (i == 0 && joff + j == 0 && k == 0) { val.x = 0.0; val.y = 0.0; }
    b[q] = val;
}

// Distributed real-data transforms without real-to-complex plans: the dense real right-hand side (Nxe, Nz, Ny) -- x fastest,
// then z, then y -- is read as a complex array Z of (Nxh = Nxe/2, Nz, Ny): column pair (2i', 2i'+1) = (re, im). After the
// local complex transform in (y, z) [or y only], the transforms A, B of the two real columns are separated with the
// Hermitian symmetry  A(m) = (Z(m) + conj Z(-m)) / 2,  B(m) = (Z(m) - conj Z(-m)) / 2i,  and only the modes
// j = 0 .. Ny/2 are kept: HALF the bytes of the reference's complex-to-complex transposes go over xGMI.
// Pack for transpose_y_to_x! (distributed_transpose.jl:25-95): send chunk q = modes j in [q*Nyc, (q+1)*Nyc), layout
// (i, jl, k) with i fastest. zmirror: the z direction was transformed too (mirror k -> (Nz-k) % Nz).
__global__ void __launch_bounds__(256) dist_pack_forward_kernel(const double2 *Z, double2 *send, int Nxl, int Nxh, int Ny, int Nyh,
                                                                int Nyc, int Nyp, int Nz, bool zmirror) {
    const int ih = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = blockIdx.z;
    if (ih >= Nxh || j >= Nyp || k >= Nz) return;
    const int q = j / Nyc, jl = j - q * Nyc;
    const long base = (((long)q * Nz + k) * Nyc + jl) * Nxl + 2 * ih;
    double2 A = make_double2(0.0, 0.0), B = A;
    if (j < Nyh) {
        const int jm = j == 0 ? 0 : Ny - j, km = (zmirror && k != 0) ? Nz - k : k;
        const double2 z1 = Z[ih + (long)Nxh * (k + (long)Nz * j)];
        const double2 z2 = Z[ih + (long)Nxh * (km + (long)Nz * jm)];
        A = make_double2(0.5 * (z1.x + z2.x), 0.5 * (z1.y - z2.y));
        B = make_double2(0.5 * (z1.y + z2.y), 0.5 * (z2.x - z1.x));
    }
    send[base] = A;
    if (2 * ih + 1 < Nxl) send[base + 1] = B;
}

// Unpack after transpose_x_to_y!: rebuild Z = A + iB for ALL modes j (upper half from the conjugate symmetry)
__global__ void __launch_bounds__(256) dist_combine_backward_kernel(const double2 *recv, double2 *Z, int Nxl, int Nxh, int Ny, int Nyh,
                                                                    int Nyc, int Nz, bool zmirror) {
    const int ih = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int k = blockIdx.z;
    if (ih >= Nxh || j >= Ny || k >= Nz) return;
    const bool cj = j >= Nyh;
    const int jj = cj ? Ny - j : j, kk = (cj && zmirror && k != 0) ? Nz - k : k;
    const int q = jj / Nyc, jl = jj - q * Nyc;
    const long base = (((long)q * Nz + kk) * Nyc + jl) * Nxl + 2 * ih;
    const double2 A = recv[base];
    const double2 B = (2 * ih + 1 < Nxl) ? recv[base + 1] : make_double2(0.0, 0.0);
    Z[ih + (long)Nxh * (k + (long)Nz * j)] = cj ? make_double2(A.x + B.y, B.x - A.y) : make_double2(A.x - B.y, A.y + B.x);
}

// real (Nxe, Nz, Ny) -> haloed pressure interior (copy_real_component!, fft_based_poisson_solver.jl:129-137)
__global__ void __launch_bounds__(256) dist_copy_real_kernel(DGrid g, FView phi, const double *src, int Nxe) {
    const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    const int j = 1 + blockIdx.y * blockDim.y + threadIdx.y;
    const int k = 1 + blockIdx.z;
    if (i > g.Nx || j > g.Ny || k > g.Nz) return;
    phi.at(i, j, k) = src[(long)(i - 1) + (long)Nxe * ((k - 1) + (long)g.Nz * (j - 1))];
}

// ---------------------------------------------------------------------------------------------------------------------
// FFT plan self-check (see ocn_api.hip: verify_*): deterministic pseudo-random pattern, round trip, compare
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double selfcheck_pattern(long q) {
    unsigned h = (unsigned)(q * 2654435761u + 12345u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (double)(h & 0xFFFFFu) / 1048576.0 - 0.5;
}
__global__ void __launch_bounds__(256) selfcheck_fill_real(double *x, long n) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) x[q] = selfcheck_pattern(q);
}
__global__ void __launch_bounds__(256) selfcheck_fill_complex(double2 *x, long n) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) x[q] = make_double2(selfcheck_pattern(q), selfcheck_pattern(q + n));
}
// max |out * scale - pattern| over the dense index space; `out` is either dense (P == N, H == 0) or the interior of a haloed array
__global__ void __launch_bounds__(256) selfcheck_compare_real(const double *out, int Nx, int Ny, int Nz, int Px, int Py, int Hx, int Hy,
                                                              int Hz, double scale, double *blockmax) {
    __shared__ double sm[256];
    double m = 0;
    const long n = (long)Nx * Ny * Nz;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        int i = q % Nx, j = (q / Nx) % Ny, k = q / ((long)Nx * Ny);
        double v = out[(long)(i + Hx) + (long)Px * ((j + Hy) + (long)Py * (k + Hz))];
        double d = fabs(v * scale - selfcheck_pattern(q));
        m = (d > m || d != d) ? d : m;
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { double o = sm[threadIdx.x + s]; if (o > sm[threadIdx.x] || o != o) sm[threadIdx.x] = o; }
        __syncthreads();
    }
    if (threadIdx.x == 0) blockmax[blockIdx.x] = sm[0];
}
__global__ void __launch_bounds__(256) selfcheck_compare_complex(const double2 *out, long n, double scale, double *blockmax) {
    __shared__ double sm[256];
    double m = 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long)gridDim.x * blockDim.x) {
        double d = fmax(fabs(out[q].x * scale - selfcheck_pattern(q)), fabs(out[q].y * scale - selfcheck_pattern(q + n)));
        m = (d > m || d != d) ? d : m;
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { double o = sm[threadIdx.x + s]; if (o > sm[threadIdx.x] || o != o) sm[threadIdx.x] = o; }
        __syncthreads();
    }
    if (threadIdx.x == 0) blockmax[blockIdx.x] = sm[0];
}
