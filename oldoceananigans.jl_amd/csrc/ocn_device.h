// ocn_device.h -- device-side building blocks of the MI355X (gfx950) NonhydrostaticModel hot path.
//
// Arithmetic contract (DESIGN.md "numerics"): this translation unit is compiled with -ffp-contract=off; a fused
// multiply-add is emitted only where the reference writes `@muladd` (src/Advection/weno_interpolants.jl:261,500,
// centered_reconstruction.jl:47-53) or `fma` (src/Utils/newton_div.jl:18) -- through __builtin_fma. Everything else is
// separate IEEE mul/add, so results are bit-comparable with the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ocn_weno_coeffs.h"

#define OCN_MAX_FIELDS 11   // u, v, w + 8 tracers

// Device view of a RectilinearGrid (reference: src/Grids/rectilinear_grid.jl:3-25) with the metric products of
// src/Operators/spacings_and_areas_and_volumes.jl precomputed per k (x, y regular): all tables are indexed [k-1+Hz],
// k = 1-Hz .. Nz+Hz+1.
struct DGrid {
    int Nx, Ny, Nz, Hx, Hy, Hz;
    int tx, ty, tz;            // 0 periodic, 1 bounded
    double dx, dy;             // Δxᶜ = Δxᶠ, Δyᶜ = Δyᶠ
    double az;                 // Azᶜᶜᶠ = Δx Δy
    double rdx, rdy;           // 1/Δxᶠ, 1/Δyᶠ
    const double *dzc, *dzf;   // Δzᵃᵃᶜ, Δzᵃᵃᶠ
    const double *ax;          // Axᶠᶜᶜ = Δy Δzᶜ
    const double *ay;          // Ayᶜᶠᶜ = Δx Δzᶜ
    const double *vinv_c;      // 1 / ((Δx Δy) Δzᶜ)   = V⁻¹ᶜᶜᶜ = V⁻¹ᶠᶜᶜ = V⁻¹ᶜᶠᶜ
    const double *vinv_f;      // 1 / ((Δx Δy) Δzᶠ)   = V⁻¹ᶜᶜᶠ
    const double *rdzf;        // 1 / Δzᶠ
    const double *rdzc;        // 1 / Δzᶜ   (the same IEEE quotient `1.0 / dzc` a kernel would form: table look-up == recomputation)
    // adapt_advection_order (Advection/adapt_advection_order.jl:18-96): buffer of the scheme a direction ends up with --
    // 3: WENO{3} (order 5), 2: WENO{2} = WENO(order = 2N-1) when N = 2, 1: UpwindBiased{1} (unused: N = 1 is refused)
    int Bx, By, Bz;
};

// A haloed field addressed with the reference's 1-based (i, j, k): p[off + i + s1*j + s2*k]
struct FView {
    double *p;
    int s1;
    long s2;
    long off;
    __device__ __forceinline__ double &at(int i, int j, int k) const { return p[off + i + (long)s1 * j + s2 * k]; }
    __device__ __forceinline__ long lin(int i, int j, int k) const { return off + i + (long)s1 * j + s2 * k; }
    template <int D> __device__ __forceinline__ long stride() const { return D == 0 ? 1L : (D == 1 ? (long)s1 : s2); }
};

struct Range6 { int i0, i1, j0, j1, k0, k1; };

// ---------------------------------------------------------------------------------------------------------------------
// WENO(order = 5) reconstruction -- src/Advection/weno_interpolants.jl
// ---------------------------------------------------------------------------------------------------------------------
// newton_div(Float32, a, b): src/Utils/newton_div.jl:8-20. The Float32 reciprocal must be the correctly rounded IEEE
// quotient (the CPU reference evaluates `inv_fast` as a true divide); compiled with
// -fhip-fp32-correctly-rounded-divide-sqrt so `1.0f / x` is exact, not v_rcp_f32.
// Correctly rounded Float32 reciprocal without the generic IEEE-divide expansion (v_div_scale / v_div_fmas /
// v_div_fixup, ~13 VALU): hardware v_rcp_f32 (1 ulp) + one Markstein correction step. Exact for every normal x whose
// reciprocal is normal -- verified EXHAUSTIVELY over all 2^23 significands by ocn_debug_rcp_check (tests/test_gpu_parity.py);
// the WENO argument beta + eps lies in [1e-8, ~1e19], far inside that range.
#ifndef OCN_RCP_VARIANT
#define OCN_RCP_VARIANT 1
#endif
template <int VARIANT> __device__ __forceinline__ float rcp_rn_f32(float x) {
    if (VARIANT == 0) return 1.0f / x;                       // compiler's correctly rounded divide
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(r, e, r);
    if (VARIANT == 2) {
        e = __builtin_fmaf(-x, r, 1.0f);
        r = __builtin_fmaf(r, e, r);
    }
    return r;
}

// Correctly rounded Float64 reciprocal for the WENO weight normalisation `1 / sum(α)` (weno_interpolants.jl:336): the compiler's
// IEEE divide expands to v_div_scale x2, v_rcp_f64, 2 Newton steps, v_mul, v_fma, v_div_fmas, v_div_fixup (11 VALU). For a
// numerator of 1 and a denominator far from the exponent limits the scale factors are 1, v_div_fmas is a plain fma and
// v_div_fixup passes its argument through, so the SAME operation sequence without them (7 VALU) returns the same bits.
// sum(α) >= C★-sum = 1 and <= ~1e45 here. Checked against `1.0 / x` on sampled inputs by ocn_debug_rcp64_check.
#ifndef OCN_RCP64_VARIANT
#define OCN_RCP64_VARIANT 1
#endif
__device__ __forceinline__ double rcp_rn_f64(double d) {
    if (OCN_RCP64_VARIANT == 0) return 1.0 / d;
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);          // n - d*q with q = n*r = r
    return __builtin_fma(e, r, r);
}

__device__ __forceinline__ double newton_div_f32(double a, double b) {
    float b_low = (float)b;
    float inv_b = rcp_rn_f32<OCN_RCP_VARIANT>(b_low);
    double inv_d = (double)inv_b;
    double x = a * inv_d;
    return __builtin_fma(__builtin_fma(x, -b, a), inv_d, x);
}

// smoothness_operation (buffer 3) under @muladd, weno_interpolants.jl:204-216,261
__device__ __forceinline__ double beta3(double p0, double p1, double p2, double C1, double C2, double C3, double C4,
                                        double C5, double C6) {
    double in1 = __builtin_fma(C3, p2, __builtin_fma(C2, p1, C1 * p0));
    double in2 = __builtin_fma(C5, p2, C4 * p1);
    return __builtin_fma(p2 * p2, C6, __builtin_fma(p1, in2, p0 * in1));
}
__device__ __forceinline__ double beta2(double p0, double p1, double C1, double C2, double C3) {
    double in1 = __builtin_fma(C2, p1, C1 * p0);
    return __builtin_fma(p1 * p1, C3, p0 * in1);
}

// ---- arithmetic modes of the flux-sharing kernels (template parameter ARITH; option "arithmetic") ----
// 0 (default, every parity test): the reference's operation sequence -- bit-identical to the oracle.
// 1 (opt-in, "contracted"): the same formulas with the roundings north_star's 1e-12 tolerance leaves free removed from the
//   instruction stream of WENO{3} -- (i) the sub-stencil polynomials `sum(coeff .* psi)` (weno_interpolants.jl:136-137) as fma chains,
//   (ii) alpha = C (1 + r^2) (:290-297) as C * fma(r, r, 1), (iii) ONE normalisation: (sum alpha_r p_r) / (sum alpha_r) instead of
//   three weights alpha_r * (1 / sum) (:336-337,500), (iv) that quotient through v_rcp_f64 + two Newton steps without the IEEE divide's
//   residual fix-up, (v) the advecting transport of the momentum fluxes interpolated before it is multiplied by the (uniform) area.
//   Smoothness indicators, tau, the Float32-reciprocal ratio with its Newton step and the upwind selection are untouched. Every change
//   is a relative perturbation of a few 2^-53 of the flux; what it does to the fields is measured, not assumed
//   (tests/test_gpu_arithmetic_mode.py).
__device__ __forceinline__ double rcp_newton2_f64(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}

// biased_interpolate for WENO{3} (weno_interpolants.jl:504-516); s0..s5 = psi[f-3 .. f+2]
template <int ARITH = 0>
__device__ __forceinline__ double weno5_biased(double s0, double s1, double s2, double s3, double s4, double s5, bool left) {
    // S₀₃, S₁₃, S₂₃ (:435-437): the right-biased stencils are the mirrored left-biased ones
    double a0 = left ? s2 : s3, a1 = left ? s3 : s2, a2 = left ? s4 : s1;   // stencil 0
    double b0 = left ? s1 : s4, b1 = a0,             b2 = a1;               // stencil 1 = (s1,s2,s3) | (s4,s3,s2)
    double c0 = left ? s0 : s5, c1 = b0,             c2 = a0;               // stencil 2 = (s0,s1,s2) | (s5,s4,s3)
    double be0 = beta3(a0, a1, a2, 10, -31, 11, 25, -19, 4);
    double be1 = beta3(b0, b1, b2, 4, -13, 5, 13, -13, 4);
    double be2 = beta3(c0, c1, c2, 4, -19, 11, 25, -31, 10);
    double tau = fabs(be0 - be2);
    double r0 = newton_div_f32(tau, be0 + OCN_WENO_EPS);
    double r1 = newton_div_f32(tau, be1 + OCN_WENO_EPS);
    double r2 = newton_div_f32(tau, be2 + OCN_WENO_EPS);
    if (ARITH == 1) {
        const double al0 = OCN_W3C0 * __builtin_fma(r0, r0, 1.0);
        const double al1 = OCN_W3C1 * __builtin_fma(r1, r1, 1.0);
        const double al2 = OCN_W3C2 * __builtin_fma(r2, r2, 1.0);
        const double q0 = __builtin_fma(OCN_W3P02, a2, __builtin_fma(OCN_W3P01, a1, OCN_W3P00 * a0));
        const double q1 = __builtin_fma(OCN_W3P12, b2, __builtin_fma(OCN_W3P11, b1, OCN_W3P10 * b0));
        const double q2 = __builtin_fma(OCN_W3P22, c2, __builtin_fma(OCN_W3P21, c1, OCN_W3P20 * c0));
        const double num = __builtin_fma(al2, q2, __builtin_fma(al1, q1, al0 * q0));
        return num * rcp_newton2_f64((al0 + al1) + al2);
    }
    double al0 = OCN_W3C0 * (1.0 + r0 * r0);
    double al1 = OCN_W3C1 * (1.0 + r1 * r1);
    double al2 = OCN_W3C2 * (1.0 + r2 * r2);
    double sinv = rcp_rn_f64((al0 + al1) + al2);
    double w0 = al0 * sinv, w1 = al1 * sinv, w2 = al2 * sinv;
    double q0 = (OCN_W3P00 * a0 + OCN_W3P01 * a1) + OCN_W3P02 * a2;
    double q1 = (OCN_W3P10 * b0 + OCN_W3P11 * b1) + OCN_W3P12 * b2;
    double q2 = (OCN_W3P20 * c0 + OCN_W3P21 * c1) + OCN_W3P22 * c2;
    return __builtin_fma(w2, q2, __builtin_fma(w1, q1, w0 * q0));
}

// WENO{2} (buffer scheme); s0..s3 = psi[f-2 .. f+1]
__device__ __forceinline__ double weno3_biased(double s0, double s1, double s2, double s3, bool left) {
    double a0 = left ? s1 : s2, a1 = left ? s2 : s1;
    double b0 = left ? s0 : s3, b1 = a0;
    double be0 = beta2(a0, a1, 1, -2, 1);
    double be1 = beta2(b0, b1, 1, -2, 1);
    double tau = fabs(be0 - be1);
    double r0 = newton_div_f32(tau, be0 + OCN_WENO_EPS);
    double r1 = newton_div_f32(tau, be1 + OCN_WENO_EPS);
    double al0 = OCN_W2C0 * (1.0 + r0 * r0);
    double al1 = OCN_W2C1 * (1.0 + r1 * r1);
    double sinv = rcp_rn_f64(al0 + al1);
    double w0 = al0 * sinv, w1 = al1 * sinv;
    double q0 = OCN_W2P00 * a0 + OCN_W2P01 * a1;
    double q1 = OCN_W2P10 * b0 + OCN_W2P11 * b1;
    return __builtin_fma(w1, q1, w0 * q0);
}

// walls of a direction by topology code (ocn_mi355x.h): Bounded has both, RightConnected the low (west) one, LeftConnected the high one
__host__ __device__ __forceinline__ bool wall_lo(int t) { return t == 1 || t == 4; }
__host__ __device__ __forceinline__ bool wall_hi(int t) { return t == 1 || t == 5; }

// topologically_conditional_interpolation.jl:46-70. `i` = index the _interpolate function is called with. Bounded tests both sides
// (:46-52); RightConnected only the left, bounded, side (:54-61); LeftConnected only the right one (:63-70).
__device__ __forceinline__ bool outside_symmetric_halo(int i, bool center, int N, int R, bool lo = true, bool hi = true) {
    const bool okl = !lo || (center ? i >= R : i >= R + 1);
    const bool okh = !hi || i <= N + 1 - R;
    return okl & okh;
}
__device__ __forceinline__ bool outside_biased_halo(int i, bool center, int N, int R, bool lo = true, bool hi = true) {
    const bool okl = !lo || (center ? ((i >= R) & (i >= R - 1)) : ((i >= R + 1) & (i >= R)));
    const bool okh = !hi || ((i <= N + 1 - (R - 1)) & (i <= N + 1 - R));
    return okl & okh;
}

// _symmetric_interpolate (scheme WENO{3} -> Centered{2}; near walls -> Centered{1}); q0..q3 = q[f-2 .. f+1]
__device__ __forceinline__ double symmetric_interp(double q0, double q1, double q2, double q3, bool bounded, int i,
                                                   bool center, int N, bool lo = true, bool hi = true) {
    bool order4 = !bounded || outside_symmetric_halo(i, center, N, 3, lo, hi);
    double c4 = __builtin_fma(OCN_C4_1, q3, __builtin_fma(OCN_C4_2, q2, __builtin_fma(OCN_C4_3, q1, OCN_C4_4 * q0)));
    if (order4) return c4;
    return __builtin_fma(OCN_C2_1, q2, OCN_C2_2 * q1);
}

// _biased_interpolate (WENO{3} -> WENO{2} -> UpwindBiased{1}); s0..s5 = psi[f-3 .. f+2]
template <int ARITH = 0>
__device__ __forceinline__ double biased_interp(double s0, double s1, double s2, double s3, double s4, double s5,
                                                bool left, bool bounded, int i, bool center, int N, bool lo = true, bool hi = true) {
    if (!bounded || outside_biased_halo(i, center, N, 3, lo, hi)) return weno5_biased<ARITH>(s0, s1, s2, s3, s4, s5, left);
    if (outside_biased_halo(i, center, N, 2, lo, hi)) return weno3_biased(s1, s2, s3, s4, left);
    return left ? 1.0 * s2 : 1.0 * s3;
}

// The same two functions for a direction whose scheme adapt_advection_order reduced to buffer B < 3 (a FluxFormAdvection,
// flux_form_advection.jl:45-59: every flux along direction d is evaluated with scheme.d -- the biased reconstruction along d AND the
// symmetric interpolation of the advecting transport along the other direction, upwind_biased_advective_fluxes.jl:23-93).
// WENO{2}: advecting_velocity_scheme Centered(order = 2), buffer_scheme UpwindBiased{1} (weno_reconstruction.jl:81-93);
// UpwindBiased{1}: Centered(order = 2), no buffer scheme (upwind_biased_reconstruction.jl:26-29).
__device__ __forceinline__ double symmetric_interp_low(double q1, double q2) { return __builtin_fma(OCN_C2_1, q2, OCN_C2_2 * q1); }
__device__ __forceinline__ double biased_interp_low(double s1, double s2, double s3, double s4, bool left, bool bounded, int i,
                                                    bool center, int N, int B, bool lo = true, bool hi = true) {
    if (B == 2 && (!bounded || outside_biased_halo(i, center, N, 2, lo, hi))) return weno3_biased(s1, s2, s3, s4, left);
    return left ? 1.0 * s2 : 1.0 * s3;
}
