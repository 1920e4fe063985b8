/* ocn_oracle.c -- CPU restatement of the Oceananigans v0.100.5 NonhydrostaticModel RK3 hot path.
 * TEST INFRASTRUCTURE ONLY (see ocn_oracle.h). Build with -ffp-contract=off: the reference fuses multiply-adds only
 * where `@muladd` is written; those sites use explicit fma() below, everything else is separate mul/add.
 * All file:line citations are relative to /root/reference/src unless stated otherwise.
 */
#include "ocn_oracle.h"
#include "../include/ocn_weno_coeffs.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double _Complex cplx;

/* ------------------------------------------------------------------------------------------------------------------
 * grid
 * ------------------------------------------------------------------------------------------------------------------ */
static double *dupvec(const double *src, int n) {
    double *p = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(p, src, sizeof(double) * (size_t)n);
    return p;
}

oro_grid *oro_grid_create(const int N[3], const int H[3], const int topo[3], const double L[3],
                          const double *dxc, const double *dxf, const double *dyc, const double *dyf,
                          const double *dzc, const double *dzf) {
    oro_grid *g = (oro_grid *)calloc(1, sizeof(oro_grid));
    const double *dc[3] = {dxc, dyc, dzc}, *df[3] = {dxf, dyf, dzf};
    for (int d = 0; d < 3; ++d) {
        g->N[d] = N[d]; g->H[d] = H[d]; g->topo[d] = topo[d]; g->L[d] = L[d];
        int len = N[d] + 2 * H[d] + 1;
        g->dc[d] = dupvec(dc[d], len);
        g->df[d] = dupvec(df[d], len);
        /* NonhydrostaticModel(advection = WENO()) adapts the scheme to the grid (nonhydrostatic_model.jl:176-179) */
        g->B[d] = oro_adapt_advection_order(ORO_ADV_WENO, 3, N[d], topo[d]);
    }
    return g;
}

/* Advection/adapt_advection_order.jl:62-96, one direction. Flat: `adapt_advection_order(::Flat, advection, N, grid) = advection`
 * (:63); Centered{B}: N >= B ? same : Centered(order = 2N) (:76-82) whose buffer is 2N / 2 = N (centered_reconstruction.jl:14);
 * UpwindBiased{B}: N >= B ? same : UpwindBiased(order = 2N-1) (:84-90), buffer (2N-1+1) / 2 = N (upwind_biased_reconstruction.jl:16);
 * WENO{B}: N >= B ? same : WENO(order = 2N-1) (:91-97), buffer N (weno_reconstruction.jl:90; order 1 is UpwindBiased(order=1), :81-83). */
int oro_adapt_advection_order(int family, int B, int N, int topo) {
    (void)family;                      /* the three families reduce to the same buffer; the family itself is kept */
    if (topo == ORO_FLAT) return B;
    return N >= B ? B : N;
}

void oro_grid_destroy(oro_grid *g) {
    if (!g) return;
    for (int d = 0; d < 3; ++d) { free(g->dc[d]); free(g->df[d]); }
    free(g);
}

/* Grids/grid_utils.jl:66-72: Face fields on Bounded dims have N+1 interior points */
void oro_parent_size(const oro_grid *g, const int loc[3], int P[3]) {
    for (int d = 0; d < 3; ++d)
        P[d] = g->N[d] + 2 * g->H[d] + ((loc[d] == ORO_FACE && g->topo[d] == ORO_BOUNDED) ? 1 : 0);
}

#define DC(g, d, i) ((g)->dc[d][(i) - 1 + (g)->H[d]])
#define DF(g, d, i) ((g)->df[d][(i) - 1 + (g)->H[d]])

/* a field view addressed with Julia's 1-based (i, j, k) */
typedef struct { double *p; long s1, s2; long off; } fld;

static fld mkfld(const oro_grid *g, const double *p, const int loc[3]) {
    int P[3];
    oro_parent_size(g, loc, P);
    fld f;
    f.p = (double *)p;
    f.s1 = P[0];
    f.s2 = (long)P[0] * P[1];
    f.off = (g->H[0] - 1) + f.s1 * (g->H[1] - 1) + f.s2 * (g->H[2] - 1);
    return f;
}
#define AT(f, i, j, k) ((f).p[(f).off + (long)(i) + (f).s1 * (long)(j) + (f).s2 * (long)(k)])

static const int LOC_U[3] = {ORO_FACE, ORO_CENTER, ORO_CENTER};
static const int LOC_V[3] = {ORO_CENTER, ORO_FACE, ORO_CENTER};
static const int LOC_W[3] = {ORO_CENTER, ORO_CENTER, ORO_FACE};
static const int LOC_C[3] = {ORO_CENTER, ORO_CENTER, ORO_CENTER};

/* ------------------------------------------------------------------------------------------------------------------
 * halo fills
 * ------------------------------------------------------------------------------------------------------------------ */
/* BoundaryConditions/fill_halo_regions_periodic.jl:5-33. Acts on the parent array, over the WHOLE parent extent of
 * the two other dimensions (fill_halo_kernels.jl:122-135). */
static void fill_periodic(const oro_grid *g, double *c, const int loc[3], int d) {
    int P[3];
    oro_parent_size(g, loc, P);
    const int H = g->H[d], N = g->N[d];
    long st[3] = {1, P[0], (long)P[0] * P[1]};
    int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
    for (int b = 0; b < P[d2]; ++b)
        for (int a = 0; a < P[d1]; ++a) {
            double *line = c + a * st[d1] + b * st[d2];
            for (int i = 1; i <= H; ++i) {          /* 1-based parent indices as in the reference */
                line[(i - 1) * st[d]] = line[(N + i - 1) * st[d]];           /* west  */
                line[(N + H + i - 1) * st[d]] = line[(H + i - 1) * st[d]];   /* east  */
            }
        }
}

/* fill_halo_regions_flux.jl:9-27 (no-flux mirror, ONE halo cell) and fill_halo_regions_open.jl:2-7 (impenetrable wall
 * value on Face fields), launched over the interior extent of the other two dims (`:xy` etc., fill_halo_kernels.jl:69-70,
 * Utils/kernel_launching.jl:211-221). */
/* getbc (boundary_condition.jl:156-164): a number, or condition[i, j] with (i, j) the interior indices along the two tangential
 * directions of side direction d, in the order x before y before z; q = (i, j, k) of the boundary-adjacent cell */
static inline double getbc(const oro_grid *g, const oro_bc *bc, int d, const int q[3]) {
    if (!bc->array) return bc->value;
    const int t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
    return bc->array[(size_t)(q[t1] - 1) + (size_t)g->N[t1] * (size_t)(q[t2] - 1)];
}

static void fill_bounded(const oro_grid *g, double *c, const int loc[3], int d, int fill_open_bcs, const oro_bc *bcs) {
    fld f = mkfld(g, c, loc);
    const int N = g->N[d];
    int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
    const oro_bc lo_src = bcs ? bcs[2 * d] : (oro_bc){ORO_BC_DEFAULT, 0.0, NULL};
    const oro_bc hi_src = bcs ? bcs[2 * d + 1] : (oro_bc){ORO_BC_DEFAULT, 0.0, NULL};
    for (int b = 1; b <= g->N[d2]; ++b)
        for (int a = 1; a <= g->N[d1]; ++a) {
            int lo[3], hi[3], ilo[3], ihi[3];
            lo[d1] = hi[d1] = ilo[d1] = ihi[d1] = a;
            lo[d2] = hi[d2] = ilo[d2] = ihi[d2] = b;
            ilo[d] = 1;
            oro_bc lo_bc = lo_src, hi_bc = hi_src;          /* the condition at this boundary point */
            lo_bc.value = getbc(g, &lo_src, d, ilo);
            hi_bc.value = getbc(g, &hi_src, d, ilo);
            if (loc[d] == ORO_CENTER) {
                lo[d] = 0; ilo[d] = 1; hi[d] = N + 1; ihi[d] = N;
                double *h0 = &AT(f, lo[0], lo[1], lo[2]), *h1 = &AT(f, hi[0], hi[1], hi[2]);
                const double c1 = AT(f, ilo[0], ilo[1], ilo[2]), cN = AT(f, ihi[0], ihi[1], ihi[2]);
                /* Δ between the first interior and the first halo point = spacing at the boundary FACE (flip(L)) */
                const double dlo = DF(g, d, 1), dhi = DF(g, d, N + 1);
                if (lo_bc.kind == ORO_BC_VALUE)         *h0 = c1 + ((c1 - lo_bc.value) / (dlo / 2)) * (-dlo);   /* :12,41-54 */
                else if (lo_bc.kind == ORO_BC_GRADIENT) *h0 = c1 + lo_bc.value * (-dlo);
                else                                    *h0 = c1;             /* Flux / default: c[0] = c[1] */
                if (hi_bc.kind == ORO_BC_VALUE)         *h1 = cN + ((hi_bc.value - cN) / (dhi / 2)) * dhi;       /* :13,56-69 */
                else if (hi_bc.kind == ORO_BC_GRADIENT) *h1 = cN + hi_bc.value * dhi;
                else                                    *h1 = cN;
            } else if (fill_open_bcs) {             /* c[1] = value; c[N+1] = value (Open; impenetrable = 0) */
                lo[d] = 1; hi[d] = N + 1;
                AT(f, lo[0], lo[1], lo[2]) = lo_bc.kind == ORO_BC_OPEN ? lo_bc.value : 0.0;
                AT(f, hi[0], hi[1], hi[2]) = hi_bc.kind == ORO_BC_OPEN ? hi_bc.value : 0.0;
            }
        }
}

/* fill_halo_regions.jl:25-36 with the ordering of boundary_condition_ordering.jl:17-46: non-periodic sides first,
 * then periodic (which also fill corners because they span the whole parent). `sortperm(bcs_array, lt=fill_first)`
 * (:42) runs an insertion sort with an `lt` that is true for every same-class pair, which REVERSES same-class
 * entries of [west_and_east, south_and_north, bottom_and_top]: the order within a class is z, y, x. */
void oro_fill_halo_regions_bcs(const oro_grid *g, double *c, const int loc[3], const oro_bc bcs[6], int fill_open_bcs) {
    for (int d = 2; d >= 0; --d)
        if (g->topo[d] == ORO_BOUNDED) fill_bounded(g, c, loc, d, fill_open_bcs, bcs);
    for (int d = 2; d >= 0; --d)
        if (g->topo[d] == ORO_PERIODIC) fill_periodic(g, c, loc, d);
}

void oro_fill_halo_regions(const oro_grid *g, double *c, const int loc[3], int fill_open_bcs) {
    oro_fill_halo_regions_bcs(g, c, loc, NULL, fill_open_bcs);
}

/* compute_flux_bcs.jl:57-163: Gc[1] += flux * A / V on the left side, Gc[N] -= flux * A / V on the right side; A is the
 * area of the boundary face (location flipped along the boundary direction), V the volume of the boundary-adjacent cell;
 * launched over the interior extent of the two other dimensions (:xy etc.) */
static inline double spacing(const oro_grid *g, int d, int face, int idx) { return face ? DF(g, d, idx) : DC(g, d, idx); }
void oro_compute_flux_bcs(const oro_grid *g, double *Gc, const int loc[3], const oro_bc bcs[6]) {
    fld G = mkfld(g, Gc, loc);
    for (int d = 0; d < 3; ++d) {            /* compute_x_bcs!, compute_y_bcs!, compute_z_bcs! */
        if (g->topo[d] != ORO_BOUNDED) continue;
        const oro_bc lo = bcs[2 * d], hi = bcs[2 * d + 1];
        if (lo.kind != ORO_BC_FLUX && hi.kind != ORO_BC_FLUX) continue;
        const int N = g->N[d];
        int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
        for (int b = 1; b <= g->N[d2]; ++b)
            for (int a = 1; a <= g->N[d1]; ++a) {
                int q[3];
                q[d1] = a; q[d2] = b;
                for (int side = 0; side < 2; ++side) {
                    const oro_bc bc = side ? hi : lo;
                    if (bc.kind != ORO_BC_FLUX) continue;
                    q[d] = side ? N : 1;
                    const int fidx = side ? N + 1 : 1;       /* boundary face index along d */
                    /* volume(i,j,k,LX,LY,LZ) = Az * Δz (spacings_and_areas_and_volumes.jl:376), Az = Δx Δy */
                    const double vol = (spacing(g, 0, loc[0], q[0]) * spacing(g, 1, loc[1], q[1])) * spacing(g, 2, loc[2], q[2]);
                    double area;
                    if (d == 0)      area = spacing(g, 1, loc[1], q[1]) * spacing(g, 2, loc[2], q[2]);                 /* Ax = Δy Δz */
                    else if (d == 1) area = spacing(g, 0, loc[0], q[0]) * spacing(g, 2, loc[2], q[2]);                 /* Ay = Δx Δz */
                    else             area = spacing(g, 0, loc[0], q[0]) * spacing(g, 1, loc[1], q[1]);                 /* Az = Δx Δy */
                    (void)fidx;      /* x / y spacings are uniform and Az does not depend on k: the flipped index only matters on curvilinear grids */
                    const double flux = getbc(g, &bc, d, q);
                    if (side) AT(G, q[0], q[1], q[2]) -= flux * area / vol;
                    else      AT(G, q[0], q[1], q[2]) += flux * area / vol;
                }
            }
    }
}

/* One side of a field-dependent Flux condition of the linear family: flux = a + b φ[i, j, k_boundary], φ a model field at the same
 * location as the field that carries the condition in the two tangential directions -- what getbc(::ContinuousBoundaryFunction)
 * (continuous_boundary_function.jl:128-161) evaluates for func(x, y, t, φ, p) = a + b φ with field_dependencies = :φ (identity
 * interpolation, boundary-normal index 1 | N). The evaporation condition of examples/ocean_wind_mixing_and_convection.jl:125-136
 * is a = 0, b = -evaporation_rate, φ = S. Applied like a valued Flux (compute_flux_bcs.jl:57-163). */
void oro_compute_linear_flux_bc(const oro_grid *g, double *Gc, const int loc[3], int side6, double a, double b, const double *dep) {
    const int d = side6 / 2, side = side6 % 2;
    if (g->topo[d] != ORO_BOUNDED) return;
    fld G = mkfld(g, Gc, loc), P = mkfld(g, dep, loc);
    const int N = g->N[d], d1 = (d + 1) % 3, d2 = (d + 2) % 3;
    for (int bb = 1; bb <= g->N[d2]; ++bb)
        for (int aa = 1; aa <= g->N[d1]; ++aa) {
            int q[3];
            q[d1] = aa; q[d2] = bb; q[d] = side ? N : 1;
            const double vol = (spacing(g, 0, loc[0], q[0]) * spacing(g, 1, loc[1], q[1])) * spacing(g, 2, loc[2], q[2]);
            double area;
            if (d == 0)      area = spacing(g, 1, loc[1], q[1]) * spacing(g, 2, loc[2], q[2]);
            else if (d == 1) area = spacing(g, 0, loc[0], q[0]) * spacing(g, 2, loc[2], q[2]);
            else             area = spacing(g, 0, loc[0], q[0]) * spacing(g, 1, loc[1], q[1]);
            const double phi = AT(P, q[0], q[1], q[2]);
            const double flux = a == 0.0 ? b * phi : a + b * phi;
            if (side) AT(G, q[0], q[1], q[2]) -= flux * area / vol;
            else      AT(G, q[0], q[1], q[2]) += flux * area / vol;
        }
}

/* ------------------------------------------------------------------------------------------------------------------
 * WENO reconstruction
 * ------------------------------------------------------------------------------------------------------------------ */
/* Utils/newton_div.jl:8-20 with inv_FT = Float32: exact Float32 reciprocal (CPU `inv_fast` is an IEEE divide), then one
 * Newton step in Float64 with two fma. */
double oro_newton_div_f32(double a, double b) {
    float b_low = (float)b;
    float inv_b = 1.0f / b_low;
    double x = a * (double)inv_b;
    x = fma(fma(x, -b, a), (double)inv_b, x);
    return x;
}

/* Advection/weno_interpolants.jl:204-216,261: smoothness_operation for buffer 3 under @muladd
 * beta = psi1*(C1 psi1 + C2 psi2 + C3 psi3) + psi2*(C4 psi2 + C5 psi3) + psi3*psi3*C6 */
static inline double beta3(const double p[3], double C1, double C2, double C3, double C4, double C5, double C6) {
    double in1 = fma(C3, p[2], fma(C2, p[1], C1 * p[0]));
    double in2 = fma(C5, p[2], C4 * p[1]);
    return fma(p[2] * p[2], C6, fma(p[1], in2, p[0] * in1));
}
/* buffer 2: beta = psi1*(C1 psi1 + C2 psi2) + psi2*psi2*C3 */
static inline double beta2(const double p[2], double C1, double C2, double C3) {
    double in1 = fma(C2, p[1], C1 * p[0]);
    return fma(p[1] * p[1], C3, p[0] * in1);
}

/* biased_interpolate for WENO{3} (weno_interpolants.jl:504-516): S = psi[i-3 .. i+2] */
double oro_weno5_biased(const double S[6], int left) {
    double p0[3], p1[3], p2[3];
    if (left) {                                     /* :435-437 */
        p0[0] = S[2]; p0[1] = S[3]; p0[2] = S[4];
        p1[0] = S[1]; p1[1] = S[2]; p1[2] = S[3];
        p2[0] = S[0]; p2[1] = S[1]; p2[2] = S[2];
    } else {
        p0[0] = S[3]; p0[1] = S[2]; p0[2] = S[1];
        p1[0] = S[4]; p1[1] = S[3]; p1[2] = S[2];
        p2[0] = S[5]; p2[1] = S[4]; p2[2] = S[3];
    }
    /* beta_loop :280-287 with smoothness_coefficients :172-174 */
    double b0 = beta3(p0, 10, -31, 11, 25, -19, 4);
    double b1 = beta3(p1, 4, -13, 5, 13, -13, 4);
    double b2 = beta3(p2, 4, -19, 11, 25, -31, 10);
    double tau = fabs(b0 - b2);                     /* :309 */
    /* zweno_alpha_loop :290-297: C*(1 + newton_div(Float32, tau, beta + eps)^2) */
    double r0 = oro_newton_div_f32(tau, b0 + OCN_WENO_EPS);
    double r1 = oro_newton_div_f32(tau, b1 + OCN_WENO_EPS);
    double r2 = oro_newton_div_f32(tau, b2 + OCN_WENO_EPS);
    double a0 = OCN_W3C0 * (1.0 + r0 * r0);
    double a1 = OCN_W3C1 * (1.0 + r1 * r1);
    double a2 = OCN_W3C2 * (1.0 + r2 * r2);
    double sinv = 1.0 / ((a0 + a1) + a2);           /* :336 */
    double w0 = a0 * sinv, w1 = a1 * sinv, w2 = a2 * sinv;
    /* biased_p :136-137: sum(coeff .* psi) -- plain products and adds */
    double q0 = (OCN_W3P00 * p0[0] + OCN_W3P01 * p0[1]) + OCN_W3P02 * p0[2];
    double q1 = (OCN_W3P10 * p1[0] + OCN_W3P11 * p1[1]) + OCN_W3P12 * p1[2];
    double q2 = (OCN_W3P20 * p2[0] + OCN_W3P21 * p2[1]) + OCN_W3P22 * p2[2];
    /* weno_reconstruction :500 under @muladd */
    return fma(w2, q2, fma(w1, q1, w0 * q0));
}

/* WENO{2} (buffer scheme of WENO{3}, weno_reconstruction.jl:77-93): S = psi[i-2 .. i+1] */
double oro_weno3_biased(const double S[4], int left) {
    double p0[2], p1[2];
    if (left) { p0[0] = S[1]; p0[1] = S[2]; p1[0] = S[0]; p1[1] = S[1]; }   /* :432-433 */
    else      { p0[0] = S[2]; p0[1] = S[1]; p1[0] = S[3]; p1[1] = S[2]; }
    double b0 = beta2(p0, 1, -2, 1);
    double b1 = beta2(p1, 1, -2, 1);
    double tau = fabs(b0 - b1);                     /* :308 */
    double r0 = oro_newton_div_f32(tau, b0 + OCN_WENO_EPS);
    double r1 = oro_newton_div_f32(tau, b1 + OCN_WENO_EPS);
    double a0 = OCN_W2C0 * (1.0 + r0 * r0);
    double a1 = OCN_W2C1 * (1.0 + r1 * r1);
    double sinv = 1.0 / (a0 + a1);
    double w0 = a0 * sinv, w1 = a1 * sinv;
    double q0 = OCN_W2P00 * p0[0] + OCN_W2P01 * p0[1];
    double q1 = OCN_W2P10 * p1[0] + OCN_W2P11 * p1[1];
    return fma(w1, q1, w0 * q0);
}

/* Advection/topologically_conditional_interpolation.jl:46-52 (Bounded) -- `i` is the index the _interpolate function is
 * called with, `center` selects the ᶜ variant (which evaluates the ᶠ stencil at i+1), R the buffer of the scheme. */
static inline int outside_symmetric_halo(int i, int center, int N, int R) {
    return center ? ((i >= R) & (i <= N + 1 - R)) : ((i >= R + 1) & (i <= N + 1 - R));
}
static inline int outside_biased_halo(int i, int center, int N, int R) {
    if (center) return (i >= R) & (i <= N + 1 - (R - 1)) & (i >= R - 1) & (i <= N + 1 - R);
    return (i >= R + 1) & (i <= N + 1 - (R - 1)) & (i >= R) & (i <= N + 1 - R);
}

/* _symmetric_interpolate for scheme WENO{3}: advecting_velocity_scheme Centered{2} in the interior
 * (upwind_biased_reconstruction.jl:52-57, centered_reconstruction.jl:45-55), falling back to the buffer schemes'
 * advecting velocity schemes (Centered{1}) near walls (topologically_conditional_interpolation.jl:93-119).
 * Q = q[f-2 .. f+1] where f is the face index of the underlying ᶠ stencil. */
static inline double symmetric_interp(const double Q[4], int bounded, int i, int center, int N) {
    int order4 = 1;
    if (bounded) order4 = outside_symmetric_halo(i, center, N, 3);
    if (order4)  /* calc_reconstruction_stencil(FT, 2, :symmetric): C4*q[-2] + C3*q[-1] + C2*q[0] + C1*q[+1] under @muladd */
        return fma(OCN_C4_1, Q[3], fma(OCN_C4_2, Q[2], fma(OCN_C4_3, Q[1], OCN_C4_4 * Q[0])));
    /* WENO{2} and UpwindBiased{1} both carry Centered(order=2): 0.5 q[-1] + 0.5 q[0] */
    return fma(OCN_C2_1, Q[2], OCN_C2_2 * Q[1]);
}

/* _biased_interpolate for scheme WENO{3} with cascade WENO{3} -> WENO{2} -> UpwindBiased{1}. S = psi[f-3 .. f+2]. */
static inline double biased_interp(const double S[6], int left, int bounded, int i, int center, int N) {
    if (!bounded || outside_biased_halo(i, center, N, 3)) return oro_weno5_biased(S, left);
    if (outside_biased_halo(i, center, N, 2)) return oro_weno3_biased(S + 1, left);
    /* UpwindBiased{1}: left -> 1.0*psi[f-1], right -> 1.0*psi[f] (calc_reconstruction_stencil buffer 1) */
    return left ? 1.0 * S[2] : 1.0 * S[3];
}

/* The same pair for a direction whose scheme adapt_advection_order reduced (FluxFormAdvection, flux_form_advection.jl:45-59):
 * WENO{2} carries advecting_velocity_scheme Centered(order=2) and buffer_scheme UpwindBiased{1} (weno_reconstruction.jl:81-93);
 * UpwindBiased{1} carries Centered(order=2) and no buffer scheme (upwind_biased_reconstruction.jl:26-29; LOADV in
 * topologically_conditional_interpolation.jl:79,98-99: never conditional). Q = q[f-1], q[f]; S = psi[f-2 .. f+1]. */
static inline double symmetric_interp_low(const double Q[2]) { return fma(OCN_C2_1, Q[1], OCN_C2_2 * Q[0]); }
static inline double biased_interp_low(const double S[4], int left, int bounded, int i, int center, int N, int B) {
    if (B == 2 && (!bounded || outside_biased_halo(i, center, N, 2))) return oro_weno3_biased(S, left);
    return left ? 1.0 * S[1] : 1.0 * S[2];
}

/* ------------------------------------------------------------------------------------------------------------------
 * advective fluxes (Advection/upwind_biased_advective_fluxes.jl:23-121)
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    const oro_grid *g;
    fld u, v, w;
} vel;

/* Operators/products_between_fields_and_grid_metrics.jl:5-14 + spacings_and_areas_and_volumes.jl:308-335 */
static inline double Ax_q_fcc(const vel *V, int i, int j, int k) { return (DC(V->g, 1, j) * DC(V->g, 2, k)) * AT(V->u, i, j, k); }
static inline double Ay_q_cfc(const vel *V, int i, int j, int k) { return (DC(V->g, 0, i) * DC(V->g, 2, k)) * AT(V->v, i, j, k); }
static inline double Az_q_ccf(const vel *V, int i, int j, int k) { return (DC(V->g, 0, i) * DC(V->g, 1, j)) * AT(V->w, i, j, k); }

typedef double (*aq_fn)(const vel *, int, int, int);

/* symmetric interpolation of an area-weighted transport along direction d. `center`: ᶜ variant (face index idx+1). */
static inline double sym_transport(const vel *V, aq_fn q, int d, int center, int i, int j, int k, int B) {
    const oro_grid *g = V->g;
    if (g->topo[d] == ORO_FLAT) return q(V, i, j, k);     /* flat_advective_fluxes.jl:35-50: interpolation along a Flat direction = ψ[i, j, k] */
    int idx = (d == 0) ? i : (d == 1) ? j : k;
    int f = idx + (center ? 1 : 0);
    if (B < 3) {   /* the scheme of the direction the flux points along was reduced: Centered(order=2) */
        double Q2[2];
        for (int n = 0; n < 2; ++n) {
            int m = f - 1 + n;
            Q2[n] = (d == 0) ? q(V, m, j, k) : (d == 1) ? q(V, i, m, k) : q(V, i, j, m);
        }
        return symmetric_interp_low(Q2);
    }
    double Q[4];
    for (int n = 0; n < 4; ++n) {
        int m = f - 2 + n;
        Q[n] = (d == 0) ? q(V, m, j, k) : (d == 1) ? q(V, i, m, k) : q(V, i, j, m);
    }
    return symmetric_interp(Q, g->topo[d] == ORO_BOUNDED, idx, center, g->N[d]);
}

static inline double biased_field(const oro_grid *g, const fld *c, int left, int d, int center, int i, int j, int k) {
    int idx = (d == 0) ? i : (d == 1) ? j : k;
    int f = idx + (center ? 1 : 0);
    if (g->B[d] < 3) {
        double S4[4] = {0, 0, 0, 0};
        for (int n = (g->B[d] == 2 ? 0 : 1); n < (g->B[d] == 2 ? 4 : 3); ++n) {
            int m = f - 2 + n;
            S4[n] = (d == 0) ? AT(*c, m, j, k) : (d == 1) ? AT(*c, i, m, k) : AT(*c, i, j, m);
        }
        return biased_interp_low(S4, left, g->topo[d] == ORO_BOUNDED, idx, center, g->N[d], g->B[d]);
    }
    double S[6];
    for (int n = 0; n < 6; ++n) {
        int m = f - 3 + n;
        S[n] = (d == 0) ? AT(*c, m, j, k) : (d == 1) ? AT(*c, i, m, k) : AT(*c, i, j, m);
    }
    return biased_interp(S, left, g->topo[d] == ORO_BOUNDED, idx, center, g->N[d]);
}

#define FLUX(name, aq, dsym, csym, dbias, cbias, fieldmember)                                                  \
    static inline double name(const vel *V, const fld *psi, int i, int j, int k) {                             \
        if (V->g->topo[dbias] == ORO_FLAT) return 0.0;   /* flat_advective_fluxes.jl:13-27 */                   \
        double ut = sym_transport(V, aq, dsym, csym, i, j, k, V->g->B[dbias]);                                 \
        double pr = biased_field(V->g, psi, ut > 0, dbias, cbias, i, j, k);                                    \
        return ut * pr;                                                                                        \
    }
/*   name      advecting  sym-dir  ᶜ?  biased-dir ᶜ? */
FLUX(flux_Uu, Ax_q_fcc, 0, 1, 0, 1, u)  /* :23-29 */
FLUX(flux_Vu, Ay_q_cfc, 0, 0, 1, 0, u)  /* :31-37 */
FLUX(flux_Wu, Az_q_ccf, 0, 0, 2, 0, u)  /* :39-45 */
FLUX(flux_Uv, Ax_q_fcc, 1, 0, 0, 0, v)  /* :47-53 */
FLUX(flux_Vv, Ay_q_cfc, 1, 1, 1, 1, v)  /* :55-61 */
FLUX(flux_Wv, Az_q_ccf, 1, 0, 2, 0, v)  /* :63-69 */
FLUX(flux_Uw, Ax_q_fcc, 2, 0, 0, 0, w)  /* :71-77 */
FLUX(flux_Vw, Ay_q_cfc, 2, 0, 1, 0, w)  /* :79-85 */
FLUX(flux_Ww, Az_q_ccf, 2, 1, 2, 1, w)  /* :87-93 */

/* tracer fluxes :99-121: Ax * u[i,j,k] * cR (left-assoc) */
static inline double flux_cx(const vel *V, const fld *c, int i, int j, int k) {
    if (V->g->topo[0] == ORO_FLAT) return 0.0;
    double ut = AT(V->u, i, j, k);
    double cr = biased_field(V->g, c, ut > 0, 0, 0, i, j, k);
    return (DC(V->g, 1, j) * DC(V->g, 2, k)) * ut * cr;
}
static inline double flux_cy(const vel *V, const fld *c, int i, int j, int k) {
    if (V->g->topo[1] == ORO_FLAT) return 0.0;
    double vt = AT(V->v, i, j, k);
    double cr = biased_field(V->g, c, vt > 0, 1, 0, i, j, k);
    return (DC(V->g, 0, i) * DC(V->g, 2, k)) * vt * cr;
}
static inline double flux_cz(const vel *V, const fld *c, int i, int j, int k) {
    if (V->g->topo[2] == ORO_FLAT) return 0.0;
    double wt = AT(V->w, i, j, k);
    double cr = biased_field(V->g, c, wt > 0, 2, 0, i, j, k);
    return (DC(V->g, 0, i) * DC(V->g, 1, j)) * wt * cr;
}

/* ------------------------------------------------------------------------------------------------------------------
 * tendencies
 * ------------------------------------------------------------------------------------------------------------------ */
/* Utils/kernel_launching.jl:145-195: exclude_periphery drops the first Face index on Bounded dims */
static void default_range(const oro_grid *g, const int loc[3], int exclude_periphery, int r[6]) {
    for (int d = 0; d < 3; ++d) {
        int o = (exclude_periphery && loc[d] == ORO_FACE && g->topo[d] == ORO_BOUNDED && g->N[d] > 1) ? 1 : 0;
        r[2 * d] = 1 + o;
        r[2 * d + 1] = g->N[d];
    }
}

static vel mkvel(const oro_grid *g, const double *u, const double *v, const double *w) {
    vel V;
    V.g = g;
    V.u = mkfld(g, u, LOC_U);
    V.v = mkfld(g, v, LOC_V);
    V.w = mkfld(g, w, LOC_W);
    return V;
}

/* nonhydrostatic_tendency_kernel_functions.jl:70-103 with every optional term `nothing`:
 * G = -div_𝐯u - 0 + 0 - 0 ... ; the chain of +/- zeros only turns -0.0 into +0.0, reproduced by `+ 0.0`. */
void oro_compute_Gu(const oro_grid *g, const double *u, const double *v, const double *w, double *Gu, const int *range) {
    int r[6];
    if (range) memcpy(r, range, sizeof r); else default_range(g, LOC_U, 1, r);
    vel V = mkvel(g, u, v, w);
    fld G = mkfld(g, Gu, LOC_U);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = r[4]; k <= r[5]; ++k)
        for (int j = r[2]; j <= r[3]; ++j)
            for (int i = r[0]; i <= r[1]; ++i) {
                /* momentum_advection_operators.jl:46-50; V⁻¹ᶠᶜᶜ = 1/(Azᶠᶜᶜ Δzᶠᶜᶜ), Az = Δxᶠ Δyᶜ */
                double Vinv = 1.0 / ((DF(g, 0, i) * DC(g, 1, j)) * DC(g, 2, k));
                double dx = flux_Uu(&V, &V.u, i, j, k) - flux_Uu(&V, &V.u, i - 1, j, k);   /* δxᶠᵃᵃ */
                double dy = flux_Vu(&V, &V.u, i, j + 1, k) - flux_Vu(&V, &V.u, i, j, k);   /* δyᵃᶜᵃ */
                double dz = flux_Wu(&V, &V.u, i, j, k + 1) - flux_Wu(&V, &V.u, i, j, k);   /* δzᵃᵃᶜ */
                double div = Vinv * ((dx + dy) + dz);
                AT(G, i, j, k) = -div + 0.0;
            }
}

void oro_compute_Gv(const oro_grid *g, const double *u, const double *v, const double *w, double *Gv, const int *range) {
    int r[6];
    if (range) memcpy(r, range, sizeof r); else default_range(g, LOC_V, 1, r);
    vel V = mkvel(g, u, v, w);
    fld G = mkfld(g, Gv, LOC_V);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = r[4]; k <= r[5]; ++k)
        for (int j = r[2]; j <= r[3]; ++j)
            for (int i = r[0]; i <= r[1]; ++i) {
                double Vinv = 1.0 / ((DC(g, 0, i) * DF(g, 1, j)) * DC(g, 2, k));            /* V⁻¹ᶜᶠᶜ */
                double dx = flux_Uv(&V, &V.v, i + 1, j, k) - flux_Uv(&V, &V.v, i, j, k);   /* δxᶜᵃᵃ */
                double dy = flux_Vv(&V, &V.v, i, j, k) - flux_Vv(&V, &V.v, i, j - 1, k);   /* δyᵃᶠᵃ */
                double dz = flux_Wv(&V, &V.v, i, j, k + 1) - flux_Wv(&V, &V.v, i, j, k);   /* δzᵃᵃᶜ */
                double div = Vinv * ((dx + dy) + dz);
                AT(G, i, j, k) = -div + 0.0;
            }
}

void oro_compute_Gw(const oro_grid *g, const double *u, const double *v, const double *w, double *Gw, const int *range) {
    int r[6];
    if (range) memcpy(r, range, sizeof r); else default_range(g, LOC_W, 1, r);
    vel V = mkvel(g, u, v, w);
    fld G = mkfld(g, Gw, LOC_W);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = r[4]; k <= r[5]; ++k)
        for (int j = r[2]; j <= r[3]; ++j)
            for (int i = r[0]; i <= r[1]; ++i) {
                double Vinv = 1.0 / ((DC(g, 0, i) * DC(g, 1, j)) * DF(g, 2, k));            /* V⁻¹ᶜᶜᶠ */
                double dx = flux_Uw(&V, &V.w, i + 1, j, k) - flux_Uw(&V, &V.w, i, j, k);   /* δxᶜᵃᵃ */
                double dy = flux_Vw(&V, &V.w, i, j + 1, k) - flux_Vw(&V, &V.w, i, j, k);   /* δyᵃᶜᵃ */
                double dz = flux_Ww(&V, &V.w, i, j, k) - flux_Ww(&V, &V.w, i, j, k - 1);   /* δzᵃᵃᶠ */
                double div = Vinv * ((dx + dy) + dz);
                AT(G, i, j, k) = -div + 0.0;
            }
}

/* tracer_advection_operators.jl:29-33; compute_Gc! is launched WITHOUT exclude_periphery (…tendencies.jl:125-127) */
void oro_compute_Gc(const oro_grid *g, const double *u, const double *v, const double *w, const double *c, double *Gc,
                    const int *range) {
    int r[6];
    if (range) memcpy(r, range, sizeof r); else default_range(g, LOC_C, 0, r);
    vel V = mkvel(g, u, v, w);
    fld C = mkfld(g, c, LOC_C);
    fld G = mkfld(g, Gc, LOC_C);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = r[4]; k <= r[5]; ++k)
        for (int j = r[2]; j <= r[3]; ++j)
            for (int i = r[0]; i <= r[1]; ++i) {
                double Vinv = 1.0 / ((DC(g, 0, i) * DC(g, 1, j)) * DC(g, 2, k));            /* V⁻¹ᶜᶜᶜ */
                double dx = flux_cx(&V, &C, i + 1, j, k) - flux_cx(&V, &C, i, j, k);
                double dy = flux_cy(&V, &C, i, j + 1, k) - flux_cy(&V, &C, i, j, k);
                double dz = flux_cz(&V, &C, i, j, k + 1) - flux_cz(&V, &C, i, j, k);
                double div = Vinv * ((dx + dy) + dz);
                AT(G, i, j, k) = -div + 0.0;
            }
}

/* ------------------------------------------------------------------------------------------------------------------
 * ScalarDiffusivity(ν, κ) -- isotropic, constant, explicit
 * TurbulenceClosures/closure_kernel_operators.jl:22-48 (flux divergences), abstract_scalar_diffusivity_closure.jl:194-242
 * (viscous_flux_* = -2 ν Σᵢⱼ, diffusive_flux_* = -κ ∂c), velocity_tracer_gradients.jl:5-42 (strain rates),
 * Operators/derivative_operators.jl:20-26 (∂ = δ * Δ⁻¹, Δ⁻¹ = 1/Δ). Differences along a Flat direction are zero.
 * Added to a tendency that already holds the advective part: G = (G - ∂ⱼτᵢⱼ) + 0.0, the order of the terms in
 * nonhydrostatic_tendency_kernel_functions.jl:91-100.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct { const oro_grid *g; fld u, v, w; double nu; int var; fld K; } visc;   /* var: coefficient = the ccc array K */
#define FLATD(g, d) ((g)->topo[d] == ORO_FLAT)
/* ∂ along d of a field: (f[idx] - f[idx-1]) at Face-in-d results, (f[idx+1] - f[idx]) at Center-in-d results */
static inline double ddx_c(const oro_grid *g, const fld *f, int i, int j, int k) { return FLATD(g, 0) ? 0.0 : (AT(*f, i + 1, j, k) - AT(*f, i, j, k)) * (1.0 / DC(g, 0, i)); }
static inline double ddy_c(const oro_grid *g, const fld *f, int i, int j, int k) { return FLATD(g, 1) ? 0.0 : (AT(*f, i, j + 1, k) - AT(*f, i, j, k)) * (1.0 / DC(g, 1, j)); }
static inline double ddz_c(const oro_grid *g, const fld *f, int i, int j, int k) { return FLATD(g, 2) ? 0.0 : (AT(*f, i, j, k + 1) - AT(*f, i, j, k)) * (1.0 / DC(g, 2, k)); }
static inline double ddx_f(const oro_grid *g, const fld *f, int i, int j, int k) { return FLATD(g, 0) ? 0.0 : (AT(*f, i, j, k) - AT(*f, i - 1, j, k)) * (1.0 / DF(g, 0, i)); }
static inline double ddy_f(const oro_grid *g, const fld *f, int i, int j, int k) { return FLATD(g, 1) ? 0.0 : (AT(*f, i, j, k) - AT(*f, i, j - 1, k)) * (1.0 / DF(g, 1, j)); }
static inline double ddz_f(const oro_grid *g, const fld *f, int i, int j, int k) { return FLATD(g, 2) ? 0.0 : (AT(*f, i, j, k) - AT(*f, i, j, k - 1)) * (1.0 / DF(g, 2, k)); }
/* strain rates (velocity_tracer_gradients.jl:25-42) */
static inline double S11(const visc *V, int i, int j, int k) { return ddx_c(V->g, &V->u, i, j, k); }
static inline double S22(const visc *V, int i, int j, int k) { return ddy_c(V->g, &V->v, i, j, k); }
static inline double S33(const visc *V, int i, int j, int k) { return ddz_c(V->g, &V->w, i, j, k); }
static inline double S12(const visc *V, int i, int j, int k) { return 0.5 * (ddy_f(V->g, &V->u, i, j, k) + ddx_f(V->g, &V->v, i, j, k)); }   /* ffc */
static inline double S13(const visc *V, int i, int j, int k) { return 0.5 * (ddz_f(V->g, &V->u, i, j, k) + ddx_f(V->g, &V->w, i, j, k)); }   /* fcf */
static inline double S23(const visc *V, int i, int j, int k) { return 0.5 * (ddz_f(V->g, &V->v, i, j, k) + ddy_f(V->g, &V->w, i, j, k)); }   /* cff */
/* A * viscous_flux: Ax_qᶜᶜᶜ(viscous_flux_ux) etc.; areas Ax = Δy Δz, Ay = Δx Δz, Az = Δx Δy at the flux location */
/* the coefficient at the flux location: a number, or a ccc array interpolated there (abstract_scalar_diffusivity_closure.jl:
 * 310-330: νᶜᶜᶜ = ν[i,j,k], νᶠᶠᶜ = ℑxyᶠᶠᵃ, νᶠᶜᶠ = ℑxzᶠᵃᶠ, νᶜᶠᶠ = ℑyzᵃᶠᶠ, κᶠᶜᶜ = ℑxᶠᵃᵃ, κᶜᶠᶜ = ℑyᵃᶠᵃ, κᶜᶜᶠ = ℑzᵃᵃᶠ; interpolation_operators.jl:8-48) */
static inline double K_ccc(const visc *V, int i, int j, int k) { return V->var ? AT(V->K, i, j, k) : V->nu; }
static inline double K_fcc(const visc *V, int i, int j, int k) { return V->var ? 0.5 * (AT(V->K, i - 1, j, k) + AT(V->K, i, j, k)) : V->nu; }
static inline double K_cfc(const visc *V, int i, int j, int k) { return V->var ? 0.5 * (AT(V->K, i, j - 1, k) + AT(V->K, i, j, k)) : V->nu; }
static inline double K_ccf(const visc *V, int i, int j, int k) { return V->var ? 0.5 * (AT(V->K, i, j, k - 1) + AT(V->K, i, j, k)) : V->nu; }
static inline double K_ffc(const visc *V, int i, int j, int k) { return V->var ? 0.5 * (K_fcc(V, i, j - 1, k) + K_fcc(V, i, j, k)) : V->nu; }   /* ℑy(ℑx) */
static inline double K_fcf(const visc *V, int i, int j, int k) { return V->var ? 0.5 * (K_fcc(V, i, j, k - 1) + K_fcc(V, i, j, k)) : V->nu; }   /* ℑz(ℑx) */
static inline double K_cff(const visc *V, int i, int j, int k) { return V->var ? 0.5 * (K_cfc(V, i, j, k - 1) + K_cfc(V, i, j, k)) : V->nu; }   /* ℑz(ℑy) */
#define VF(L, S) (-(2 * (K_##L(V, i, j, k) * (S))))
static inline double AxFux(const visc *V, int i, int j, int k) { return (DC(V->g, 1, j) * DC(V->g, 2, k)) * VF(ccc, S11(V, i, j, k)); }   /* ccc */
static inline double AyFuy(const visc *V, int i, int j, int k) { return (DF(V->g, 0, i) * DC(V->g, 2, k)) * VF(ffc, S12(V, i, j, k)); }   /* ffc */
static inline double AzFuz(const visc *V, int i, int j, int k) { return (DF(V->g, 0, i) * DC(V->g, 1, j)) * VF(fcf, S13(V, i, j, k)); }   /* fcf */
static inline double AxFvx(const visc *V, int i, int j, int k) { return (DF(V->g, 1, j) * DC(V->g, 2, k)) * VF(ffc, S12(V, i, j, k)); }   /* ffc */
static inline double AyFvy(const visc *V, int i, int j, int k) { return (DC(V->g, 0, i) * DC(V->g, 2, k)) * VF(ccc, S22(V, i, j, k)); }   /* ccc */
static inline double AzFvz(const visc *V, int i, int j, int k) { return (DC(V->g, 0, i) * DF(V->g, 1, j)) * VF(cff, S23(V, i, j, k)); }   /* cff */
static inline double AxFwx(const visc *V, int i, int j, int k) { return (DC(V->g, 1, j) * DF(V->g, 2, k)) * VF(fcf, S13(V, i, j, k)); }   /* fcf */
static inline double AyFwy(const visc *V, int i, int j, int k) { return (DC(V->g, 0, i) * DF(V->g, 2, k)) * VF(cff, S23(V, i, j, k)); }   /* cff */
static inline double AzFwz(const visc *V, int i, int j, int k) { return (DC(V->g, 0, i) * DC(V->g, 1, j)) * VF(ccc, S33(V, i, j, k)); }   /* ccc */
#undef VF

/* which: 0 u, 1 v, 2 w (coef = ν), 3 tracer c (coef = κ) */
static void add_closure_tendency(const oro_grid *g, int which, const double *u, const double *v, const double *w, const double *c,
                                 double coef, const double *coef_ccc, double *Gp, const int *range);

void oro_add_closure_tendency(const oro_grid *g, int which, const double *u, const double *v, const double *w, const double *c,
                              double coef, double *Gp, const int *range) {
    add_closure_tendency(g, which, u, v, w, c, coef, NULL, Gp, range);
}

/* the same with the coefficient read from a ccc array with filled halos (eddy viscosity / diffusivity of an LES closure) */
void oro_add_closure_tendency_field(const oro_grid *g, int which, const double *u, const double *v, const double *w, const double *c,
                                    const double *coef_ccc, double *Gp, const int *range) {
    add_closure_tendency(g, which, u, v, w, c, 0.0, coef_ccc, Gp, range);
}

static void add_closure_tendency(const oro_grid *g, int which, const double *u, const double *v, const double *w, const double *c,
                                 double coef, const double *coef_ccc, double *Gp, const int *range) {
    static const int *LOCS[4] = {LOC_U, LOC_V, LOC_W, LOC_C};
    int r[6];
    if (range) memcpy(r, range, sizeof r); else default_range(g, LOCS[which], which < 3, r);
    visc Vs = {g, mkfld(g, u, LOC_U), mkfld(g, v, LOC_V), mkfld(g, w, LOC_W), coef, coef_ccc != NULL, mkfld(g, coef_ccc ? coef_ccc : u, LOC_C)};
    const visc *V = &Vs;
    fld C = mkfld(g, c ? c : u, LOC_C);
    fld G = mkfld(g, Gp, LOCS[which]);
    const int fx = FLATD(g, 0), fy = FLATD(g, 1), fz = FLATD(g, 2);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = r[4]; k <= r[5]; ++k)
        for (int j = r[2]; j <= r[3]; ++j)
            for (int i = r[0]; i <= r[1]; ++i) {
                double dx, dy, dz, Vinv;
                if (which == 0) {            /* ∂ⱼ_τ₁ⱼ at fcc */
                    Vinv = 1.0 / ((DF(g, 0, i) * DC(g, 1, j)) * DC(g, 2, k));
                    dx = fx ? 0.0 : AxFux(V, i, j, k) - AxFux(V, i - 1, j, k);
                    dy = fy ? 0.0 : AyFuy(V, i, j + 1, k) - AyFuy(V, i, j, k);
                    dz = fz ? 0.0 : AzFuz(V, i, j, k + 1) - AzFuz(V, i, j, k);
                } else if (which == 1) {     /* ∂ⱼ_τ₂ⱼ at cfc */
                    Vinv = 1.0 / ((DC(g, 0, i) * DF(g, 1, j)) * DC(g, 2, k));
                    dx = fx ? 0.0 : AxFvx(V, i + 1, j, k) - AxFvx(V, i, j, k);
                    dy = fy ? 0.0 : AyFvy(V, i, j, k) - AyFvy(V, i, j - 1, k);
                    dz = fz ? 0.0 : AzFvz(V, i, j, k + 1) - AzFvz(V, i, j, k);
                } else if (which == 2) {     /* ∂ⱼ_τ₃ⱼ at ccf */
                    Vinv = 1.0 / ((DC(g, 0, i) * DC(g, 1, j)) * DF(g, 2, k));
                    dx = fx ? 0.0 : AxFwx(V, i + 1, j, k) - AxFwx(V, i, j, k);
                    dy = fy ? 0.0 : AyFwy(V, i, j + 1, k) - AyFwy(V, i, j, k);
                    dz = fz ? 0.0 : AzFwz(V, i, j, k) - AzFwz(V, i, j, k - 1);
                } else {                     /* ∇_dot_qᶜ at ccc: A * (-(κ ∂c)) */
                    Vinv = 1.0 / ((DC(g, 0, i) * DC(g, 1, j)) * DC(g, 2, k));
                    const double ax = DC(g, 1, j) * DC(g, 2, k), ay = DC(g, 0, i) * DC(g, 2, k), az = DC(g, 0, i) * DC(g, 1, j);
                    dx = fx ? 0.0 : ax * -(K_fcc(V, i + 1, j, k) * ddx_f(g, &C, i + 1, j, k)) - ax * -(K_fcc(V, i, j, k) * ddx_f(g, &C, i, j, k));
                    dy = fy ? 0.0 : ay * -(K_cfc(V, i, j + 1, k) * ddy_f(g, &C, i, j + 1, k)) - ay * -(K_cfc(V, i, j, k) * ddy_f(g, &C, i, j, k));
                    dz = fz ? 0.0 : az * -(K_ccf(V, i, j, k + 1) * ddz_f(g, &C, i, j, k + 1)) - az * -(K_ccf(V, i, j, k) * ddz_f(g, &C, i, j, k));
                }
                const double div = Vinv * ((dx + dy) + dz);
                AT(G, i, j, k) = (AT(G, i, j, k) - div) + 0.0;
            }
}

/* ------------------------------------------------------------------------------------------------------------------
 * AnisotropicMinimumDissipation(C, Cν, Cκ; Cb = nothing) -- the eddy viscosity νₑ and the eddy diffusivities κₑ
 * at ccc: turbulence_closure_implementations/anisotropic_minimum_dissipation.jl:152-196 (the two kernels), :226-357 (filter
 * widths = 2Δ at the CALLING index, "the 30 terms", tr ∇u, the tracer terms -- including the ℑxzᶜᵃᶜ of norm_∂y_w at :323, kept
 * as written) and velocity_tracer_gradients.jl:116-250 (norm_∂x_u = ∂x_u unscaled, norm_∂x_v = Δᶠx/Δᶠy ∂x_v, ...).
 * n-ary + and * fold left to right, x^2 is x*x, unary minus binds before *, Cb = nothing makes (r - Cb_ζ) = r - 0.
 * PARITY UNPINNED by reference data: the reference's tests only check that a model with this closure time-steps
 * (test_time_stepping.jl:257,400); tests/test_oracle_kats.py pins the restatement on flows with known answers instead
 * (laminar shear -> 0; u = (x, y, -2z) -> νₑ = Cν δ², κₑ = 2 Cκ δ² for c = z).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct { const oro_grid *g; fld u, v, w, c; } amd;
typedef double (*amd_op)(const amd *, int, int, int);
#define FX(A, i) (2 * DC((A)->g, 0, i))     /* Δᶠx = 2 Δxᶜᶜᶜ(i, j, k); _ffc, _fcf, ... are aliases evaluated at the same index */
#define FY(A, j) (2 * DC((A)->g, 1, j))
#define FZ(A, k) (2 * DC((A)->g, 2, k))
static double n_dxu(const amd *A, int i, int j, int k) { return ddx_c(A->g, &A->u, i, j, k); }
static double n_dyv(const amd *A, int i, int j, int k) { return ddy_c(A->g, &A->v, i, j, k); }
static double n_dzw(const amd *A, int i, int j, int k) { return ddz_c(A->g, &A->w, i, j, k); }
static double n_dxv(const amd *A, int i, int j, int k) { return FX(A, i) / FY(A, j) * ddx_f(A->g, &A->v, i, j, k); }   /* ffc */
static double n_dyu(const amd *A, int i, int j, int k) { return FY(A, j) / FX(A, i) * ddy_f(A->g, &A->u, i, j, k); }   /* ffc */
static double n_dxw(const amd *A, int i, int j, int k) { return FX(A, i) / FZ(A, k) * ddx_f(A->g, &A->w, i, j, k); }   /* fcf */
static double n_dzu(const amd *A, int i, int j, int k) { return FZ(A, k) / FX(A, i) * ddz_f(A->g, &A->u, i, j, k); }   /* fcf */
static double n_dyw(const amd *A, int i, int j, int k) { return FY(A, j) / FZ(A, k) * ddy_f(A->g, &A->w, i, j, k); }   /* cff */
static double n_dzv(const amd *A, int i, int j, int k) { return FZ(A, k) / FY(A, j) * ddz_f(A->g, &A->v, i, j, k); }   /* cff */
static double n_S12(const amd *A, int i, int j, int k) { return 0.5 * (n_dyu(A, i, j, k) + n_dxv(A, i, j, k)); }
static double n_S13(const amd *A, int i, int j, int k) { return 0.5 * (n_dzu(A, i, j, k) + n_dxw(A, i, j, k)); }
static double n_S23(const amd *A, int i, int j, int k) { return 0.5 * (n_dzv(A, i, j, k) + n_dyw(A, i, j, k)); }
#define SQ(name, f) static double name(const amd *A, int i, int j, int k) { const double x = f(A, i, j, k); return x * x; }
#define PR(name, f, h) static double name(const amd *A, int i, int j, int k) { return f(A, i, j, k) * h(A, i, j, k); }
SQ(n_dxv2, n_dxv) SQ(n_dyu2, n_dyu) SQ(n_dxw2, n_dxw) SQ(n_dzu2, n_dzu) SQ(n_dyw2, n_dyw) SQ(n_dzv2, n_dzv)
PR(n_dxv_S12, n_dxv, n_S12) PR(n_dyu_S12, n_dyu, n_S12) PR(n_dxw_S13, n_dxw, n_S13) PR(n_dzu_S13, n_dzu, n_S13)
PR(n_dzv_S23, n_dzv, n_S23) PR(n_dyw_S23, n_dyw, n_S23)
static double n_dxc(const amd *A, int i, int j, int k) { return FX(A, i) * ddx_f(A->g, &A->c, i, j, k); }   /* fcc */
static double n_dyc(const amd *A, int i, int j, int k) { return FY(A, j) * ddy_f(A->g, &A->c, i, j, k); }   /* cfc */
static double n_dzc(const amd *A, int i, int j, int k) { return FZ(A, k) * ddz_f(A->g, &A->c, i, j, k); }   /* ccf */
SQ(n_dxc2, n_dxc) SQ(n_dyc2, n_dyc) SQ(n_dzc2, n_dzc)
#undef SQ
#undef PR
/* ℑxᶜᵃᵃ, ℑyᵃᶜᵃ, ℑzᵃᵃᶜ of a function and the double interpolations built from them (interpolation_operators.jl:20-48) */
static inline double Ix(const amd *A, amd_op f, int i, int j, int k) { return 0.5 * (f(A, i, j, k) + f(A, i + 1, j, k)); }
static inline double Iy(const amd *A, amd_op f, int i, int j, int k) { return 0.5 * (f(A, i, j, k) + f(A, i, j + 1, k)); }
static inline double Iz(const amd *A, amd_op f, int i, int j, int k) { return 0.5 * (f(A, i, j, k) + f(A, i, j, k + 1)); }
static inline double Ixy(const amd *A, amd_op f, int i, int j, int k) { return 0.5 * (Ix(A, f, i, j, k) + Ix(A, f, i, j + 1, k)); }
static inline double Ixz(const amd *A, amd_op f, int i, int j, int k) { return 0.5 * (Ix(A, f, i, j, k) + Ix(A, f, i, j, k + 1)); }
static inline double Iyz(const amd *A, amd_op f, int i, int j, int k) { return 0.5 * (Iy(A, f, i, j, k) + Iy(A, f, i, j, k + 1)); }

static double amd_delta2(const amd *A, int i, int j, int k) {
    const double fx = FX(A, i), fy = FY(A, j), fz = FZ(A, k);
    return 3 / ((1 / (fx * fx) + 1 / (fy * fy)) + 1 / (fz * fz));
}

static double amd_viscosity(const amd *A, double Cnu, int i, int j, int k) {
    const double dxu = n_dxu(A, i, j, k), dyv = n_dyv(A, i, j, k), dzw = n_dzw(A, i, j, k);
    /* norm_tr_∇uᶜᶜᶜ (:275-297) */
    double q = dxu * dxu + dyv * dyv;
    q = q + dzw * dzw;
    q = q + Ixy(A, n_dxv2, i, j, k);
    q = q + Ixy(A, n_dyu2, i, j, k);
    q = q + Ixz(A, n_dxw2, i, j, k);
    q = q + Ixz(A, n_dzu2, i, j, k);
    q = q + Iyz(A, n_dyw2, i, j, k);
    q = q + Iyz(A, n_dzv2, i, j, k);
    if (q == 0) return fmax(0.0, 0.0);
    /* norm_uᵢₐ_uⱼₐ_Σᵢⱼᶜᶜᶜ (:226-269); norm_Σ₁₁ = norm_∂x_u etc. */
    double b1 = dxu * (dxu * dxu) + dyv * Ixy(A, n_dxv2, i, j, k);
    b1 = b1 + dzw * Ixz(A, n_dxw2, i, j, k);
    b1 = b1 + 2 * dxu * Ixy(A, n_dxv_S12, i, j, k);
    b1 = b1 + 2 * dxu * Ixz(A, n_dxw_S13, i, j, k);
    b1 = b1 + 2 * Ixy(A, n_dxv, i, j, k) * Ixz(A, n_dxw, i, j, k) * Iyz(A, n_S23, i, j, k);
    double b2 = dxu * Ixy(A, n_dyu2, i, j, k) + dyv * (dyv * dyv);
    b2 = b2 + dzw * Iyz(A, n_dyw2, i, j, k);
    b2 = b2 + 2 * dyv * Ixy(A, n_dyu_S12, i, j, k);
    b2 = b2 + 2 * Ixy(A, n_dyu, i, j, k) * Iyz(A, n_dyw, i, j, k) * Ixz(A, n_S13, i, j, k);
    b2 = b2 + 2 * dyv * Iyz(A, n_dyw_S23, i, j, k);
    double b3 = dxu * Ixz(A, n_dzu2, i, j, k) + dyv * Iyz(A, n_dzv2, i, j, k);
    b3 = b3 + dzw * (dzw * dzw);
    b3 = b3 + 2 * Ixz(A, n_dzu, i, j, k) * Iyz(A, n_dzv, i, j, k) * Ixy(A, n_S12, i, j, k);
    b3 = b3 + 2 * dzw * Ixz(A, n_dzu_S13, i, j, k);
    b3 = b3 + 2 * dzw * Iyz(A, n_dzv_S23, i, j, k);
    const double r = (b1 + b2) + b3;
    const double Cb_zeta = 0.0 / FZ(A, k);                              /* Cb = nothing */
    const double nu = -Cnu * amd_delta2(A, i, j, k) * (r - Cb_zeta) / q;
    return fmax(0.0, nu);
}

static double amd_diffusivity(const amd *A, double Ck, int i, int j, int k) {
    const double sigma = (Ix(A, n_dxc2, i, j, k) + Iy(A, n_dyc2, i, j, k)) + Iz(A, n_dzc2, i, j, k);   /* norm_θᵢ²ᶜᶜᶜ (:331-333) */
    if (sigma == 0) return fmax(0.0, 0.0);
    const double dxu = n_dxu(A, i, j, k), dyv = n_dyv(A, i, j, k), dzw = n_dzw(A, i, j, k);
    const double cx = Ix(A, n_dxc, i, j, k), cy = Iy(A, n_dyc, i, j, k), cz = Iz(A, n_dzc, i, j, k);
    /* norm_uᵢⱼ_cⱼ_cᵢᶜᶜᶜ (:306-329) */
    double t1 = dxu * Ix(A, n_dxc2, i, j, k) + Ixy(A, n_dxv, i, j, k) * cx * cy;
    t1 = t1 + Ixz(A, n_dxw, i, j, k) * cx * cz;
    double t2 = Ixy(A, n_dyu, i, j, k) * cy * cx + dyv * Iy(A, n_dyc2, i, j, k);
    t2 = t2 + Ixz(A, n_dyw, i, j, k) * cy * cz;                        /* ℑxzᶜᵃᶜ of a cff quantity: as the reference has it */
    double t3 = Ixz(A, n_dzu, i, j, k) * cz * cx + Iyz(A, n_dzv, i, j, k) * cz * cy;
    t3 = t3 + dzw * Iz(A, n_dzc2, i, j, k);
    const double theta = (t1 + t2) + t3;
    const double kap = -Ck * amd_delta2(A, i, j, k) * theta / sigma;
    return fmax(0.0, kap);
}
#undef FX
#undef FY
#undef FZ

/* compute_diffusivities!(..., closure::AMD, model; parameters = :xyz) (:199-216): νₑ and one κₑ per tracer over the interior.
 * The caller fills their halos (fill_halo_regions!(diffusivity_fields; only_local_halos = true), default ccc conditions). */
void oro_compute_amd_diffusivities(const oro_grid *g, double Cnu, const double *Ckappa, const double *u, const double *v,
                                   const double *w, const double *const *tracers, int ntracers, double *nu_e, double *const *kappa_e) {
    amd As = {g, mkfld(g, u, LOC_U), mkfld(g, v, LOC_V), mkfld(g, w, LOC_W), mkfld(g, u, LOC_C)};
    fld NU = mkfld(g, nu_e, LOC_C);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->N[2]; ++k)
        for (int j = 1; j <= g->N[1]; ++j)
            for (int i = 1; i <= g->N[0]; ++i) AT(NU, i, j, k) = amd_viscosity(&As, Cnu, i, j, k);
    for (int t = 0; t < ntracers; ++t) {
        amd At = As;
        At.c = mkfld(g, tracers[t], LOC_C);
        fld KA = mkfld(g, kappa_e[t], LOC_C);
#pragma omp parallel for collapse(2) schedule(static)
        for (int k = 1; k <= g->N[2]; ++k)
            for (int j = 1; j <= g->N[1]; ++j)
                for (int i = 1; i <= g->N[0]; ++i) AT(KA, i, j, k) = amd_diffusivity(&At, Ckappa[t], i, j, k);
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Buoyancy: BuoyancyTracer and SeawaterBuoyancy with a LinearEquationOfState, gravity along -z.
 * NonhydrostaticModel separates the hydrostatic pressure anomaly pHY′ whenever buoyancy is present
 * (nonhydrostatic_model.jl:144-158): update_hydrostatic_pressure! (update_hydrostatic_pressure.jl:12-22) integrates
 * z_dot_g_bᶜᶜᶠ = ĝ_z ℑzᵃᵃᶠ(b) (g_dot_b.jl:4, ĝ_z = 1) downwards over i = 0:Nx+1, j = 0:Ny+1 (:43-50); u and v tendencies get
 * -∂x pHY′, -∂y pHY′ (nonhydrostatic_tendency_kernel_functions.jl:14-19,97,159), the w tendency gets no buoyancy term (:168).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct { int kind; const double *b, *T, *S; double g, alpha, beta; } buoy;   /* kind 1: tracer b; 2: linear seawater */
static inline double buoyancy_perturbation(const oro_grid *g, const buoy *B, int i, int j, int k) {
    if (B->kind == 1) { fld b = mkfld(g, B->b, LOC_C); return AT(b, i, j, k); }                 /* buoyancy_tracer.jl:12 */
    fld T = mkfld(g, B->T, LOC_C), S = mkfld(g, B->S, LOC_C);                                   /* linear_equation_of_state.jl:71-73 */
    return B->g * (B->alpha * AT(T, i, j, k) - B->beta * AT(S, i, j, k));
}

void oro_update_hydrostatic_pressure(const oro_grid *g, int kind, const double *b_or_T, const double *S, double grav, double alpha,
                                     double beta, double *pHY) {
    if (g->topo[2] == ORO_FLAT) return;                               /* update_hydrostatic_pressure!(::ZFlatGrid) = nothing */
    const buoy B = {kind, b_or_T, b_or_T, S, grav, alpha, beta};
    fld P = mkfld(g, pHY, LOC_C);
    const int Nz = g->N[2];
    const int i0 = g->topo[0] == ORO_FLAT ? 1 : 0, i1 = g->topo[0] == ORO_FLAT ? g->N[0] : g->N[0] + 1;
    const int j0 = g->topo[1] == ORO_FLAT ? 1 : 0, j1 = g->topo[1] == ORO_FLAT ? g->N[1] : g->N[1] + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
            /* z_dot_g_bᶜᶜᶠ(k) = 1 * (0.5 * (b[k-1] + b[k])) */
            double zb = 1 * (0.5 * (buoyancy_perturbation(g, &B, i, j, Nz) + buoyancy_perturbation(g, &B, i, j, Nz + 1)));
            AT(P, i, j, Nz) = -zb * DF(g, 2, Nz + 1);
            for (int k = Nz - 1; k >= 1; --k) {
                zb = 1 * (0.5 * (buoyancy_perturbation(g, &B, i, j, k) + buoyancy_perturbation(g, &B, i, j, k + 1)));
                AT(P, i, j, k) = AT(P, i, j, k + 1) - zb * DF(g, 2, k + 1);
            }
        }
}

/* G_u -= ∂xᶠᶜᶜ pHY′, G_v -= ∂yᶜᶠᶜ pHY′ on tendencies holding the advective part (the terms between are zeros) */
void oro_add_hydrostatic_pressure_gradient(const oro_grid *g, const double *pHY, double *Gu, double *Gv) {
    fld P = mkfld(g, pHY, LOC_C), GU = mkfld(g, Gu, LOC_U), GV = mkfld(g, Gv, LOC_V);
    int ru[6], rv[6];
    default_range(g, LOC_U, 1, ru);
    default_range(g, LOC_V, 1, rv);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->N[2]; ++k)
        for (int j = 1; j <= g->N[1]; ++j)
            for (int i = 1; i <= g->N[0]; ++i) {
                if (i >= ru[0] && j >= ru[2])
                    AT(GU, i, j, k) = AT(GU, i, j, k) - (g->topo[0] == ORO_FLAT ? 0.0 : (AT(P, i, j, k) - AT(P, i - 1, j, k)) * (1.0 / DF(g, 0, i)));
                if (i >= rv[0] && j >= rv[2])
                    AT(GV, i, j, k) = AT(GV, i, j, k) - (g->topo[1] == ORO_FLAT ? 0.0 : (AT(P, i, j, k) - AT(P, i, j - 1, k)) * (1.0 / DF(g, 1, j)));
            }
}

/* ------------------------------------------------------------------------------------------------------------------
 * coriolis = FPlane(f) (Coriolis/f_plane.jl:48-52): x_f_cross_U = -f * active_weighted_ℑxyᶠᶜᶜ(v),
 * y_f_cross_U = f * active_weighted_ℑxyᶜᶠᶜ(u), z = 0. The active-weighted average (Operators/interpolation_operators.jl:116-130)
 * divides the four-point average by the fraction of nodes that are not peripheral (Grids/inactive_node.jl:152-156: a node is
 * peripheral when one of the cells it touches lies outside a Bounded direction); interpolation along a Flat direction is the identity.
 * Added to tendencies holding the advective part: G_u = G_u - x_f_cross_U, G_v = G_v - y_f_cross_U.
 * ------------------------------------------------------------------------------------------------------------------ */
static inline int inactive_cell(const oro_grid *g, int i, int j, int k) {
    return (g->topo[0] == ORO_BOUNDED && (i < 1 || i > g->N[0])) | (g->topo[1] == ORO_BOUNDED && (j < 1 || j > g->N[1])) |
           (g->topo[2] == ORO_BOUNDED && (k < 1 || k > g->N[2]));
}
/* not_peripheral_node at (c, f, c) [v nodes] and (f, c, c) [u nodes], as a Float64 0 / 1 */
static inline double np_cfc(const oro_grid *g, int i, int j, int k) { return !(inactive_cell(g, i, j, k) | inactive_cell(g, i, j - 1, k)) ? 1.0 : 0.0; }
static inline double np_fcc(const oro_grid *g, int i, int j, int k) { return !(inactive_cell(g, i, j, k) | inactive_cell(g, i - 1, j, k)) ? 1.0 : 0.0; }

void oro_add_fplane_coriolis(const oro_grid *g, double f, const double *u, const double *v, double *Gu, double *Gv) {
    fld U = mkfld(g, u, LOC_U), V = mkfld(g, v, LOC_V), GU = mkfld(g, Gu, LOC_U), GV = mkfld(g, Gv, LOC_V);
    const int fx = g->topo[0] == ORO_FLAT, fy = g->topo[1] == ORO_FLAT;
    int ru[6], rv[6];
    default_range(g, LOC_U, 1, ru);
    default_range(g, LOC_V, 1, rv);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->N[2]; ++k)
        for (int j = 1; j <= g->N[1]; ++j)
            for (int i = 1; i <= g->N[0]; ++i) {
                if (i >= ru[0] && j >= ru[2]) {
                    /* ℑxyᶠᶜᵃ(q) = ℑyᵃᶜᵃ(ℑxᶠᵃᵃ q): 0.5 * (X(j) + X(j+1)), X(j) = 0.5 * (q[i-1, j] + q[i, j]) */
#define XF(F, jj) (fx ? F(i, jj) : 0.5 * (F(i - 1, jj) + F(i, jj)))
#define VV(ii, jj) AT(V, ii, jj, k)
#define NPV(ii, jj) np_cfc(g, ii, jj, k)
                    const double qa = fy ? XF(VV, j) : 0.5 * (XF(VV, j) + XF(VV, j + 1));
                    const double an = fy ? XF(NPV, j) : 0.5 * (XF(NPV, j) + XF(NPV, j + 1));
                    const double aw = an == 0 ? 0.0 : qa / an;
                    AT(GU, i, j, k) = AT(GU, i, j, k) - (-f * aw);
#undef XF
#undef VV
#undef NPV
                }
                if (i >= rv[0] && j >= rv[2]) {
                    /* ℑxyᶜᶠᵃ(q) = ℑyᵃᶠᵃ(ℑxᶜᵃᵃ q): 0.5 * (X(j-1) + X(j)), X(j) = 0.5 * (q[i, j] + q[i+1, j]) */
#define XC(F, jj) (fx ? F(i, jj) : 0.5 * (F(i, jj) + F(i + 1, jj)))
#define UU(ii, jj) AT(U, ii, jj, k)
#define NPU(ii, jj) np_fcc(g, ii, jj, k)
                    const double qa = fy ? XC(UU, j) : 0.5 * (XC(UU, j - 1) + XC(UU, j));
                    const double an = fy ? XC(NPU, j) : 0.5 * (XC(NPU, j - 1) + XC(NPU, j));
                    const double aw = an == 0 ? 0.0 : qa / an;
                    AT(GV, i, j, k) = AT(GV, i, j, k) - (f * aw);
#undef XC
#undef UU
#undef NPU
                }
            }
}

/* ------------------------------------------------------------------------------------------------------------------
 * RK3 substep and tendency caching
 * ------------------------------------------------------------------------------------------------------------------ */
/* TimeSteppers/runge_kutta_3.jl:212-226, launched :xyz with exclude_periphery=true (:187) */
void oro_rk3_substep_field(const oro_grid *g, double *U, const int loc[3], double dt, double gamma, double zeta,
                           int has_zeta, const double *Gn, const double *Gm) {
    int r[6];
    default_range(g, loc, 1, r);
    fld u = mkfld(g, U, loc), gn = mkfld(g, Gn, loc), gm = mkfld(g, Gm, loc);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = r[4]; k <= r[5]; ++k)
        for (int j = r[2]; j <= r[3]; ++j)
            for (int i = r[0]; i <= r[1]; ++i) {
                if (has_zeta) AT(u, i, j, k) += dt * (gamma * AT(gn, i, j, k) + zeta * AT(gm, i, j, k));
                else          AT(u, i, j, k) += dt * gamma * AT(gn, i, j, k);
            }
}

/* TimeSteppers/store_tendencies.jl:6-22, launched :xyz (no periphery exclusion) */
void oro_cache_tendencies(const oro_grid *g, double *Gm, const double *Gn, const int loc[3]) {
    fld a = mkfld(g, Gm, loc), b = mkfld(g, Gn, loc);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->N[2]; ++k)
        for (int j = 1; j <= g->N[1]; ++j)
            for (int i = 1; i <= g->N[0]; ++i) AT(a, i, j, k) = AT(b, i, j, k);
}

/* ------------------------------------------------------------------------------------------------------------------
 * pressure source term / correction
 * ------------------------------------------------------------------------------------------------------------------ */
/* solve_for_pressure.jl:12-18 / :36-42 + Operators/divergence_operators.jl:16-19 */
void oro_compute_source_term(const oro_grid *g, const double *u, const double *v, const double *w, cplx *rhs,
                             int weight_by_dz) {
    vel V = mkvel(g, u, v, w);
    const int Nx = g->N[0], Ny = g->N[1], Nz = g->N[2];
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= Nz; ++k)
        for (int j = 1; j <= Ny; ++j)
            for (int i = 1; i <= Nx; ++i) {
                double Vinv = 1.0 / ((DC(g, 0, i) * DC(g, 1, j)) * DC(g, 2, k));
                /* δ along a Flat direction is zero(FT) (Operators/difference_operators.jl:30-49) */
                double dx = g->topo[0] == ORO_FLAT ? 0.0 : Ax_q_fcc(&V, i + 1, j, k) - Ax_q_fcc(&V, i, j, k);
                double dy = g->topo[1] == ORO_FLAT ? 0.0 : Ay_q_cfc(&V, i, j + 1, k) - Ay_q_cfc(&V, i, j, k);
                double dz = g->topo[2] == ORO_FLAT ? 0.0 : Az_q_ccf(&V, i, j, k + 1) - Az_q_ccf(&V, i, j, k);
                double div = Vinv * ((dx + dy) + dz);
                /* `active * δ` with active = true; Fourier-tridiagonal: active * Δzᶜᶜᶜ * δ */
                double val = weight_by_dz ? (1.0 * DC(g, 2, k)) * div : 1.0 * div;
                rhs[(i - 1) + (size_t)Nx * ((j - 1) + (size_t)Ny * (k - 1))] = val;
            }
}

/* pressure_correction.jl:31-37 with ∂ = δ * Δ⁻¹, Δ⁻¹ = 1/Δ (derivative_operators.jl:20-26, reciprocal_metric_operators.jl) */
void oro_make_pressure_correction(const oro_grid *g, double *u, double *v, double *w, const double *p) {
    fld U = mkfld(g, u, LOC_U), Vv = mkfld(g, v, LOC_V), W = mkfld(g, w, LOC_W), P = mkfld(g, p, LOC_C);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->N[2]; ++k)
        for (int j = 1; j <= g->N[1]; ++j)
            for (int i = 1; i <= g->N[0]; ++i) {
                AT(U, i, j, k) -= (g->topo[0] == ORO_FLAT ? 0.0 : AT(P, i, j, k) - AT(P, i - 1, j, k)) * (1.0 / DF(g, 0, i));
                AT(Vv, i, j, k) -= (g->topo[1] == ORO_FLAT ? 0.0 : AT(P, i, j, k) - AT(P, i, j - 1, k)) * (1.0 / DF(g, 1, j));
                AT(W, i, j, k) -= (g->topo[2] == ORO_FLAT ? 0.0 : AT(P, i, j, k) - AT(P, i, j, k - 1)) * (1.0 / DF(g, 2, k));
            }
}

/* pressure_correction.jl:48-50: `pNHS ./= Δt⁺` broadcasts over the field's interior... */
void oro_scale_parent(const oro_grid *g, double *p, const int loc[3], double divisor) {
    /* Field broadcasting (Fields/broadcasting_abstract_fields.jl) acts on interior(p) for a full field */
    fld P = mkfld(g, p, loc);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->N[2]; ++k)
        for (int j = 1; j <= g->N[1]; ++j)
            for (int i = 1; i <= g->N[0]; ++i) AT(P, i, j, k) /= divisor;
}

/* ------------------------------------------------------------------------------------------------------------------
 * FFT (own implementation: iterative radix-2 + Bluestein for other lengths) and FFTW-convention DCTs
 * The reference calls FFTW (Solvers/plan_transforms.jl:16-34); bitwise FFT parity is unpinned by the reference.
 * ------------------------------------------------------------------------------------------------------------------ */
/* twiddle cache: w[k] = exp(-2 pi i k / n), k < n/2, built once per length (not thread safe to build: built
 * serially before the parallel line loops via fft_prepare) */
#define ORO_MAX_TW 64
static struct { int n; cplx *w; } g_tw[ORO_MAX_TW];
static int g_ntw = 0;
static const cplx *twiddles(int n) {
    for (int q = 0; q < g_ntw; ++q) if (g_tw[q].n == n) return g_tw[q].w;
    if (g_ntw == ORO_MAX_TW) g_ntw = 0;
    cplx *w = (cplx *)malloc(sizeof(cplx) * (size_t)(n / 2 + 1));
    for (int k = 0; k < n / 2; ++k) {
        double ang = -2.0 * M_PI * (double)k / (double)n;
        w[k] = cos(ang) + I * sin(ang);
    }
    g_tw[g_ntw].n = n; g_tw[g_ntw].w = w;
    return g_tw[g_ntw++].w;
}

static void fft_pow2(cplx *x, int n, int sign) {
    const cplx *w = twiddles(n);
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { cplx t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < half; ++k) {
                cplx wk = sign < 0 ? w[k * step] : conj(w[k * step]);
                cplx a = x[i + k], b = x[i + k + half] * wk;
                x[i + k] = a + b;
                x[i + k + half] = a - b;
            }
    }
}

static int bluestein_len(int n) {
    int m = 1;
    while (m < 2 * n - 1) m <<= 1;
    return m;
}

static void fft_prepare(int n) {            /* call serially before threaded use */
    if (n <= 1) return;
    if ((n & (n - 1)) == 0) (void)twiddles(n);
    else (void)twiddles(bluestein_len(n));
}

static void fft_any(cplx *x, int n, int sign) {
    if (n <= 1) return;
    if ((n & (n - 1)) == 0) { fft_pow2(x, n, sign); return; }
    /* Bluestein */
    int m = bluestein_len(n);
    cplx *a = (cplx *)calloc((size_t)m, sizeof(cplx)), *b = (cplx *)calloc((size_t)m, sizeof(cplx));
    cplx *wv = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    for (int k = 0; k < n; ++k) {
        long kk = ((long)k * k) % (2L * n);
        double ang = sign * M_PI * (double)kk / (double)n;
        wv[k] = cos(ang) + I * sin(ang);
    }
    for (int k = 0; k < n; ++k) a[k] = x[k] * wv[k];
    b[0] = conj(wv[0]);
    for (int k = 1; k < n; ++k) b[k] = b[m - k] = conj(wv[k]);
    fft_pow2(a, m, -1);
    fft_pow2(b, m, -1);
    for (int k = 0; k < m; ++k) a[k] *= b[k];
    fft_pow2(a, m, +1);
    for (int k = 0; k < n; ++k) x[k] = (a[k] / (double)m) * wv[k];
    free(a); free(b); free(wv);
}

void oro_fft_line(cplx *x, int n, int stride, int sign) {
    fft_prepare(n);
    cplx stackbuf[512];
    cplx *tmp = n <= 512 ? stackbuf : (cplx *)malloc(sizeof(cplx) * (size_t)n);
    for (int i = 0; i < n; ++i) tmp[i] = x[(size_t)i * stride];
    fft_any(tmp, n, sign);
    for (int i = 0; i < n; ++i) x[(size_t)i * stride] = tmp[i];
    if (tmp != stackbuf) free(tmp);
}

/* FFTW REDFT10: Y_k = 2 sum_j X_j cos(pi (j+1/2) k / n); REDFT01: Y_k = X_0 + 2 sum_{j>=1} X_j cos(pi j (k+1/2) / n).
 * Applied to real and imaginary parts independently (FFTW r2r on a complex array acts on both components). */
static void dct_line(cplx *x, int n, int stride, int backward) {
    cplx *tmp = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    for (int k = 0; k < n; ++k) {
        cplx s = 0;
        if (!backward) {
            for (int j = 0; j < n; ++j) s += x[(size_t)j * stride] * cos(M_PI * (j + 0.5) * k / n);
            tmp[k] = 2.0 * s;
        } else {
            for (int j = 1; j < n; ++j) s += x[(size_t)j * stride] * cos(M_PI * j * (k + 0.5) / n);
            tmp[k] = x[0] + 2.0 * s;
        }
    }
    for (int k = 0; k < n; ++k) x[(size_t)k * stride] = tmp[k];
    free(tmp);
}

/* Solvers/index_permutations.jl:18-35 (Makhoul 1980 eq. 20), 1-based */
int oro_permute_index(int i, int N) { return (i % 2 == 1) ? (int)(i / 2.0) + 1 : N - (int)((i - 1) / 2.0); }
int oro_unpermute_index(int i, int N) { return (i <= ceil(N / 2.0)) ? 2 * i - 1 : 2 * (N - i + 1); }

/* The transforms as the CPU DiscreteTransform applies them (FFTW REDFT10 forward; REDFT01 x 1/2N backward,
 * discrete_transforms.jl:20-34) on n real numbers */
void oro_dct_direct(double *x, int n, int backward) {
    cplx *t = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    for (int k = 0; k < n; ++k) t[k] = x[k];
    dct_line(t, n, 1, backward);
    for (int k = 0; k < n; ++k) x[k] = creal(t[k]) * (backward ? 1.0 / (2.0 * n) : 1.0);
    free(t);
}

/* ... and as the GPU DiscreteTransform applies them (discrete_transforms.jl:108-175): forward = permute_indices!, complex FFT,
 * A = 2 real(omega_4N^k A) (:166-169, twiddles :48-75: omega(M, k) = exp(-2 pi i k / M), Solvers.jl); backward = A *= omega_4N^-k with
 * the zeroth factor halved, normalised inverse FFT, unpermute_indices!, real part */
void oro_dct_makhoul(double *x, int n, int backward) {
    cplx *A = (cplx *)malloc(sizeof(cplx) * (size_t)n), *Bf = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    fft_prepare(n);
    if (!backward) {
        for (int i = 1; i <= n; ++i) Bf[oro_permute_index(i, n) - 1] = x[i - 1];
        fft_any(Bf, n, -1);
        for (int k = 0; k < n; ++k) {
            double ang = -2.0 * M_PI * (double)k / (4.0 * n);
            x[k] = 2.0 * creal((cos(ang) + I * sin(ang)) * Bf[k]);
        }
    } else {
        for (int k = 0; k < n; ++k) {
            double ang = -2.0 * M_PI * (double)(-k) / (4.0 * n);
            A[k] = x[k] * (cos(ang) + I * sin(ang)) * (k == 0 ? 0.5 : 1.0);
        }
        fft_any(A, n, +1);
        for (int i = 1; i <= n; ++i) Bf[oro_unpermute_index(i, n) - 1] = A[i - 1] / (double)n;
        for (int k = 0; k < n; ++k) x[k] = creal(Bf[k]);
    }
    free(A); free(Bf);
}

/* transform all lines along dimension d of a dense (n0, n1, n2) column-major complex array */
static void transform_dim(cplx *A, const int n[3], int d, int kind /*0 fft fwd,1 fft bwd(unnormalised),2 dct fwd,3 dct bwd*/) {
    size_t st[3] = {1, (size_t)n[0], (size_t)n[0] * n[1]};
    int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
    fft_prepare(n[d]);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n[d2]; ++b)
        for (int a = 0; a < n[d1]; ++a) {
            cplx *line = A + a * st[d1] + b * st[d2];
            if (kind == 0) oro_fft_line(line, n[d], (int)st[d], -1);
            else if (kind == 1) oro_fft_line(line, n[d], (int)st[d], +1);
            else dct_line(line, n[d], (int)st[d], kind == 3);
        }
}

/* Solvers/poisson_eigenvalues.jl:8-23 */
void oro_poisson_eigenvalues(int N, double L, int topo, double *lam) {
    if (topo == ORO_FLAT) { for (int i = 0; i < N; ++i) lam[i] = 0.0; return; }     /* poisson_eigenvalues.jl: Flat -> zeros */
    for (int i = 1; i <= N; ++i) {
        double arg = (topo == ORO_PERIODIC) ? ((double)(i - 1) * M_PI) / (double)N
                                            : ((double)(i - 1) * M_PI) / (double)(2 * N);
        double s = 2.0 * sin(arg) / (L / (double)N);
        lam[i - 1] = s * s;
    }
}

struct oro_poisson {
    const oro_grid *g;
    int kind;
    int n[3];
    cplx *storage;      /* solution storage (both kinds) */
    cplx *source;       /* kind 1: source_term */
    double *lam[3];
    double *D, *lower, *t;   /* kind 1: main diagonal (3-D), lower/upper diagonal, scratch */
};

oro_poisson *oro_poisson_create(const oro_grid *g, int kind) {
    oro_poisson *s = (oro_poisson *)calloc(1, sizeof(oro_poisson));
    s->g = g; s->kind = kind;
    size_t tot = 1;
    for (int d = 0; d < 3; ++d) { s->n[d] = g->N[d]; tot *= (size_t)g->N[d]; }
    s->storage = (cplx *)calloc(tot, sizeof(cplx));
    for (int d = 0; d < 3; ++d) {
        s->lam[d] = (double *)calloc((size_t)g->N[d], sizeof(double));
        oro_poisson_eigenvalues(g->N[d], g->L[d], g->topo[d], s->lam[d]);
    }
    if (kind == 1) {
        /* fourier_tridiagonal_poisson_solver.jl:75-134, diagonals :180-210 (HomogeneousZFormulation) */
        const int Nx = g->N[0], Ny = g->N[1], Nz = g->N[2];
        s->source = (cplx *)calloc(tot, sizeof(cplx));
        s->D = (double *)calloc(tot, sizeof(double));
        s->t = (double *)calloc(tot, sizeof(double));
        s->lower = (double *)calloc((size_t)(Nz > 1 ? Nz - 1 : 1), sizeof(double));
        for (int j = 1; j <= Ny; ++j)
            for (int i = 1; i <= Nx; ++i) {
                double lxy = s->lam[0][i - 1] + s->lam[1][j - 1];
#define DD(k) s->D[(i - 1) + (size_t)Nx * ((j - 1) + (size_t)Ny * ((k) - 1))]
                DD(1) = -1.0 / DF(g, 2, 2) - DC(g, 2, 1) * lxy;
                DD(Nz) = -1.0 / DF(g, 2, Nz) - DC(g, 2, Nz) * lxy;
                for (int k = 2; k <= Nz - 1; ++k)
                    DD(k) = -(1.0 / DF(g, 2, k + 1) + 1.0 / DF(g, 2, k)) - DC(g, 2, k) * lxy;
#undef DD
            }
        for (int q = 1; q <= Nz - 1; ++q) s->lower[q - 1] = 1.0 / DF(g, 2, q + 1);
    }
    return s;
}

void oro_poisson_destroy(oro_poisson *s) {
    if (!s) return;
    free(s->storage); free(s->source); free(s->D); free(s->lower); free(s->t);
    for (int d = 0; d < 3; ++d) free(s->lam[d]);
    free(s);
}

cplx *oro_poisson_rhs(oro_poisson *s) { return s->kind == 0 ? s->storage : s->source; }

/* Solvers/batched_tridiagonal_solver.jl:213-245 (z direction): complex f/phi, real a, b (3-D), c; scratch t */
void oro_batched_tridiagonal_solve_z(int Nx, int Ny, int Nz, const double *a, const double *b3d, const double *c,
                                     const cplx *f, double *t, cplx *phi) {
#define IX(i, j, k) ((size_t)(i) + (size_t)Nx * ((size_t)(j) + (size_t)Ny * (size_t)(k)))
#pragma omp parallel for collapse(2) schedule(static)
    for (int j = 0; j < Ny; ++j)
        for (int i = 0; i < Nx; ++i) {
            double beta = b3d[IX(i, j, 0)];
            phi[IX(i, j, 0)] = f[IX(i, j, 0)] / beta;
            for (int k = 1; k < Nz; ++k) {
                double ck1 = c[k - 1], bk = b3d[IX(i, j, k)], ak1 = a[k - 1];
                t[IX(i, j, k)] = ck1 / beta;
                beta = bk - ak1 * t[IX(i, j, k)];
                cplx fk = f[IX(i, j, k)];
                int dd = fabs(beta) > 10.0 * 2.220446049250313e-16;
                cplx star = (fk - ak1 * phi[IX(i, j, k - 1)]) / beta;
                if (dd) phi[IX(i, j, k)] = star;    /* else keep the previous content of phi (:236-237) */
            }
            for (int k = Nz - 2; k >= 0; --k) phi[IX(i, j, k)] -= t[IX(i, j, k + 1)] * phi[IX(i, j, k + 1)];
        }
#undef IX
}

static void copy_real_component(const oro_grid *g, double *phi, const cplx *src) {
    /* fft_based_poisson_solver.jl:129-137 */
    fld P = mkfld(g, phi, LOC_C);
    const int Nx = g->N[0], Ny = g->N[1];
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k <= g->N[2]; ++k)
        for (int j = 1; j <= Ny; ++j)
            for (int i = 1; i <= Nx; ++i)
                AT(P, i, j, k) = creal(src[(i - 1) + (size_t)Nx * ((j - 1) + (size_t)Ny * (k - 1))]);
}

void oro_poisson_solve(oro_poisson *s, double *phi) {
    const oro_grid *g = s->g;
    const int *n = s->n;
    const size_t tot = (size_t)n[0] * n[1] * n[2];
    if (s->kind == 0) {
        /* fft_based_poisson_solver.jl:95-125; CPU plans: forward = (bounded dims REDFT10, periodic dims FFT),
         * backward = (periodic IFFT (normalised), bounded REDFT01 * 1/2N) -- plan_transforms.jl:124-136 */
        cplx *b = s->storage;
        for (int d = 0; d < 3; ++d) if (g->topo[d] == ORO_BOUNDED) transform_dim(b, n, d, 2);
        for (int d = 0; d < 3; ++d) if (g->topo[d] == ORO_PERIODIC) transform_dim(b, n, d, 0);
#pragma omp parallel for collapse(2) schedule(static)
        for (int k = 0; k < n[2]; ++k)
            for (int j = 0; j < n[1]; ++j)
                for (int i = 0; i < n[0]; ++i) {
                    size_t q = (size_t)i + (size_t)n[0] * ((size_t)j + (size_t)n[1] * k);
                    double lam = (s->lam[0][i] + s->lam[1][j]) + s->lam[2][k] - 0.0;
                    b[q] = (-creal(b[q]) / lam) + I * (-cimag(b[q]) / lam);
                }
        b[0] = 0;                                                    /* :115 */
        double norm = 1.0;
        for (int d = 0; d < 3; ++d) if (g->topo[d] == ORO_PERIODIC) { transform_dim(b, n, d, 1); norm *= (double)n[d]; }
        if (norm != 1.0) {
            double inv = 1.0 / norm;
            for (size_t q = 0; q < tot; ++q) b[q] *= inv;
        }
        double bn = 1.0;
        int anyb = 0;
        for (int d = 0; d < 3; ++d) if (g->topo[d] == ORO_BOUNDED) { transform_dim(b, n, d, 3); bn *= 1.0 / (2.0 * n[d]); anyb = 1; }
        if (anyb) for (size_t q = 0; q < tot; ++q) b[q] *= bn;
        copy_real_component(g, phi, b);
    } else {
        /* fourier_tridiagonal_poisson_solver.jl:212-239 (x, y transformed; z tridiagonal) */
        cplx *src = s->source, *ph = s->storage;
        for (int d = 0; d < 2; ++d) if (g->topo[d] == ORO_BOUNDED) transform_dim(src, n, d, 2);
        for (int d = 0; d < 2; ++d) if (g->topo[d] == ORO_PERIODIC) transform_dim(src, n, d, 0);
        oro_batched_tridiagonal_solve_z(n[0], n[1], n[2], s->lower, s->D, s->lower, src, s->t, ph);
        double norm = 1.0;
        for (int d = 0; d < 2; ++d) if (g->topo[d] == ORO_PERIODIC) { transform_dim(ph, n, d, 1); norm *= (double)n[d]; }
        if (norm != 1.0) {
            double inv = 1.0 / norm;
            for (size_t q = 0; q < tot; ++q) ph[q] *= inv;
        }
        double bn = 1.0;
        int anyb = 0;
        for (int d = 0; d < 2; ++d) if (g->topo[d] == ORO_BOUNDED) { transform_dim(ph, n, d, 3); bn *= 1.0 / (2.0 * n[d]); anyb = 1; }
        if (anyb) for (size_t q = 0; q < tot; ++q) ph[q] *= bn;
        /* ϕ .= ϕ .- mean(ϕ) (:233) */
        cplx sum = 0;
        for (size_t q = 0; q < tot; ++q) sum += ph[q];
        cplx mean = sum / (double)tot;
        for (size_t q = 0; q < tot; ++q) ph[q] -= mean;
        copy_real_component(g, phi, ph);
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * model + time_step!
 * ------------------------------------------------------------------------------------------------------------------ */
#define ORO_MAXTR 8
struct oro_model {
    const oro_grid *g;
    int ntr;
    double *U[3 + ORO_MAXTR];      /* u, v, w, tracers */
    double *Gn[3 + ORO_MAXTR];
    double *Gm[3 + ORO_MAXTR];
    int loc[3 + ORO_MAXTR][3];
    oro_bc bcs[3 + ORO_MAXTR][6];
    oro_bc kbcs[1 + ORO_MAXTR][6];   /* conditions of the diffusivity fields: [0] = nu_e, [1 + t] = kappa_e of tracer t */
    int any_flux_bc;
    struct { int on, dep; double a, b; } lin[3 + ORO_MAXTR][6];   /* linear field-dependent Flux conditions */
    int has_closure;
    double nu, kappa[ORO_MAXTR];
    int has_amd;                                    /* closure = AnisotropicMinimumDissipation(Cν, Cκ) */
    double Cnu, Ckappa[ORO_MAXTR];
    double *nu_e, *kappa_e[ORO_MAXTR];              /* diffusivity_fields.νₑ, .κₑ (ccc, with halos) */
    int has_coriolis;
    double fcor;
    int buoyancy_kind, b_index, T_index, S_index;   /* 0 none, 1 BuoyancyTracer, 2 linear SeawaterBuoyancy */
    double grav, alpha, beta;
    double *pHY;
    double *p;
    oro_poisson *solver;
    double time, last_dt, last_stage_dt;
    int iteration, stage;
};

static size_t parent_len(const oro_grid *g, const int loc[3]) {
    int P[3];
    oro_parent_size(g, loc, P);
    return (size_t)P[0] * P[1] * P[2];
}

static int grid_z_is_regular(const oro_grid *g) {
    for (int k = 1 - g->H[2]; k <= g->N[2] + g->H[2]; ++k)
        if (DC(g, 2, k) != DC(g, 2, 1)) return 0;
    return 1;
}

oro_model *oro_model_create(const oro_grid *g, int ntracers) {
    oro_model *m = (oro_model *)calloc(1, sizeof(oro_model));
    m->g = g; m->ntr = ntracers;
    const int *locs[3] = {LOC_U, LOC_V, LOC_W};
    for (int f = 0; f < 3 + ntracers; ++f) {
        const int *l = f < 3 ? locs[f] : LOC_C;
        memcpy(m->loc[f], l, sizeof(int) * 3);
        size_t len = parent_len(g, l);
        m->U[f] = (double *)calloc(len, sizeof(double));
        m->Gn[f] = (double *)calloc(len, sizeof(double));
        m->Gm[f] = (double *)calloc(len, sizeof(double));
    }
    m->p = (double *)calloc(parent_len(g, LOC_C), sizeof(double));
    /* NonhydrostaticModels.jl:25-40: XYZ-regular -> FFTBased, z-stretched -> FourierTridiagonal */
    m->solver = oro_poisson_create(g, grid_z_is_regular(g) ? 0 : 1);
    m->stage = 1;
    m->last_dt = INFINITY; m->last_stage_dt = INFINITY;
    return m;
}

void oro_model_destroy(oro_model *m) {
    if (!m) return;
    for (int f = 0; f < 3 + m->ntr; ++f) { free(m->U[f]); free(m->Gn[f]); free(m->Gm[f]); }
    free(m->p);
    free(m->pHY);
    free(m->nu_e);
    for (int t = 0; t < m->ntr; ++t) free(m->kappa_e[t]);
    oro_poisson_destroy(m->solver);
    free(m);
}

static int field_index(const oro_model *m, const char *name, char *kind) {
    /* "u","v","w","cN","p","Gu","Gv","Gw","GcN","Mu","Mv","Mw","McN" */
    const char *q = name;
    *kind = 'U';
    if (q[0] == 'G' || q[0] == 'M') { *kind = q[0]; ++q; }
    if (!strcmp(q, "u")) return 0;
    if (!strcmp(q, "v")) return 1;
    if (!strcmp(q, "w")) return 2;
    if (q[0] == 'c') { int n = atoi(q + 1); if (n >= 0 && n < m->ntr) return 3 + n; }
    if (!strcmp(name, "p") || !strcmp(name, "pHY")) { *kind = 'p'; return 0; }
    return -1;
}

double *oro_model_field(oro_model *m, const char *name) {
    char kind;
    int f = field_index(m, name, &kind);
    if (!strcmp(name, "pHY")) return m->pHY;
    if (!strcmp(name, "nu_e")) return m->nu_e;
    if (!strncmp(name, "kappa_e", 7) && name[7] >= '0' && name[7] - '0' < m->ntr) return m->kappa_e[name[7] - '0'];
    if (f < 0) return NULL;
    if (kind == 'p') return m->p;
    return kind == 'U' ? m->U[f] : kind == 'G' ? m->Gn[f] : m->Gm[f];
}

void oro_model_field_loc(oro_model *m, const char *name, int loc[3]) {
    char kind;
    int f = field_index(m, name, &kind);
    if (kind == 'p' || f < 0) { memcpy(loc, LOC_C, sizeof(int) * 3); return; }
    memcpy(loc, m->loc[f], sizeof(int) * 3);
}

/* FieldBoundaryConditions validation (field_boundary_conditions.jl, boundary_condition.jl): Flux / Value / Gradient on
 * fields at Center along the boundary direction, Open on the wall-normal (Face) component, Bounded sides only */
static int model_set_bc(oro_model *m, const char *name, int side, int kind, double value, const double *array);
int oro_model_set_bc(oro_model *m, const char *name, int side, int kind, double value) { return model_set_bc(m, name, side, kind, value, NULL); }
/* array-valued condition: `array` is borrowed and must outlive the model's use of it */
int oro_model_set_bc_array(oro_model *m, const char *name, int side, int kind, const double *array) {
    return array ? model_set_bc(m, name, side, kind, 0.0, array) : -1;
}
static int model_set_bc(oro_model *m, const char *name, int side, int kind, double value, const double *array) {
    if (!strcmp(name, "nu_e") || (!strncmp(name, "kappa_e", 7) && name[7] >= '0' && name[7] <= '9' && !name[8])) {
        /* diffusivity fields are CenterFields with user conditions (anisotropic_minimum_dissipation.jl:339-352) */
        const int q = name[0] == 'n' ? 0 : 1 + (name[7] - '0');
        if (q > m->ntr || side < 0 || side > 5 || m->g->topo[side / 2] != ORO_BOUNDED) return -1;
        if (kind != ORO_BC_DEFAULT && kind != ORO_BC_VALUE && kind != ORO_BC_GRADIENT) return -1;
        m->kbcs[q][side].kind = kind; m->kbcs[q][side].value = value; m->kbcs[q][side].array = array;
        return 0;
    }
    char k;
    int f = field_index(m, name, &k);
    if (f < 0 || k != 'U' || side < 0 || side > 5) return -1;
    const int d = side / 2;
    if (m->g->topo[d] != ORO_BOUNDED) return -1;
    if (kind == ORO_BC_OPEN ? m->loc[f][d] != ORO_FACE : (kind != ORO_BC_DEFAULT && m->loc[f][d] != ORO_CENTER)) return -1;
    if (kind < ORO_BC_DEFAULT || kind > ORO_BC_OPEN) return -1;
    m->bcs[f][side].kind = kind;
    m->bcs[f][side].value = value;
    m->bcs[f][side].array = array;
    m->any_flux_bc = 0;
    for (int q = 0; q < 3 + m->ntr; ++q)
        for (int sd = 0; sd < 6; ++sd)
            if (m->bcs[q][sd].kind == ORO_BC_FLUX && (m->bcs[q][sd].value != 0.0 || m->bcs[q][sd].array)) m->any_flux_bc = 1;
    return 0;
}

/* buoyancy = BuoyancyTracer() (kind 1, tracer index b) | SeawaterBuoyancy(equation_of_state = LinearEquationOfState(α, β),
 * gravitational_acceleration = g) (kind 2, tracer indices T, S) | nothing (kind 0) */
int oro_model_set_buoyancy(oro_model *m, int kind, int b_or_T_index, int S_index, double grav, double alpha, double beta) {
    if (kind < 0 || kind > 2) return -1;
    if (kind && (b_or_T_index < 0 || b_or_T_index >= m->ntr)) return -1;
    if (kind == 2 && (S_index < 0 || S_index >= m->ntr)) return -1;
    m->buoyancy_kind = kind; m->b_index = m->T_index = b_or_T_index; m->S_index = S_index;
    m->grav = grav; m->alpha = alpha; m->beta = beta;
    if (kind && !m->pHY) m->pHY = (double *)calloc(parent_len(m->g, LOC_C), sizeof(double));
    return 0;
}

void oro_model_set_coriolis(oro_model *m, int has, double f) { m->has_coriolis = has; m->fcor = f; }

void oro_model_set_closure(oro_model *m, double nu, const double *kappa) {
    m->nu = nu;
    m->has_closure = nu != 0.0;
    for (int t = 0; t < m->ntr; ++t) {
        m->kappa[t] = kappa ? kappa[t] : 0.0;
        if (m->kappa[t] != 0.0) m->has_closure = 1;
    }
}

/* closure = AnisotropicMinimumDissipation(Cν = Cnu, Cκ = Ckappa[tracer]; Cb = nothing); replaces a ScalarDiffusivity */
int oro_model_set_amd(oro_model *m, double Cnu, const double *Ckappa) {
    for (int d = 0; d < 3; ++d)
        if (m->g->topo[d] == ORO_FLAT) return -1;
    m->has_amd = 1; m->has_closure = 0; m->nu = 0.0;
    m->Cnu = Cnu;
    if (!m->nu_e) m->nu_e = (double *)calloc(parent_len(m->g, LOC_C), sizeof(double));
    for (int t = 0; t < m->ntr; ++t) {
        m->kappa[t] = 0.0;
        m->Ckappa[t] = Ckappa[t];
        if (!m->kappa_e[t]) m->kappa_e[t] = (double *)calloc(parent_len(m->g, LOC_C), sizeof(double));
    }
    return 0;
}

/* update_nonhydrostatic_model_state.jl:20-56 with buoyancy = nothing */
void oro_model_update_state(oro_model *m, int compute_tendencies) {
    const oro_grid *g = m->g;
    for (int f = 0; f < 3 + m->ntr; ++f) oro_fill_halo_regions_bcs(g, m->U[f], m->loc[f], m->bcs[f], /*fill_open_bcs=*/0);
    /* compute_auxiliaries!: compute_diffusivities! over :xyz, update_hydrostatic_pressure! (update_nonhydrostatic_model_state.jl:
     * 58-69); then fill_halo_regions!(model.diffusivity_fields; only_local_halos = true) (:44) with the default ccc conditions */
    if (m->has_amd) {
        oro_compute_amd_diffusivities(g, m->Cnu, m->Ckappa, m->U[0], m->U[1], m->U[2], (const double *const *)(m->U + 3), m->ntr,
                                      m->nu_e, m->kappa_e);
        oro_fill_halo_regions_bcs(g, m->nu_e, LOC_C, m->kbcs[0], 1);
        for (int t = 0; t < m->ntr; ++t) oro_fill_halo_regions_bcs(g, m->kappa_e[t], LOC_C, m->kbcs[1 + t], 1);
    }
    if (m->buoyancy_kind == 1)
        oro_update_hydrostatic_pressure(g, 1, m->U[3 + m->b_index], NULL, 0, 0, 0, m->pHY);
    else if (m->buoyancy_kind == 2)
        oro_update_hydrostatic_pressure(g, 2, m->U[3 + m->T_index], m->U[3 + m->S_index], m->grav, m->alpha, m->beta, m->pHY);
    if (compute_tendencies) {
        oro_compute_Gu(g, m->U[0], m->U[1], m->U[2], m->Gn[0], NULL);
        oro_compute_Gv(g, m->U[0], m->U[1], m->U[2], m->Gn[1], NULL);
        oro_compute_Gw(g, m->U[0], m->U[1], m->U[2], m->Gn[2], NULL);
        for (int t = 0; t < m->ntr; ++t) oro_compute_Gc(g, m->U[0], m->U[1], m->U[2], m->U[3 + t], m->Gn[3 + t], NULL);
        if (m->has_coriolis) oro_add_fplane_coriolis(g, m->fcor, m->U[0], m->U[1], m->Gn[0], m->Gn[1]);
        if (m->buoyancy_kind) oro_add_hydrostatic_pressure_gradient(g, m->pHY, m->Gn[0], m->Gn[1]);
        if (m->has_closure) {
            for (int f = 0; f < 3; ++f) oro_add_closure_tendency(g, f, m->U[0], m->U[1], m->U[2], NULL, m->nu, m->Gn[f], NULL);
            for (int t = 0; t < m->ntr; ++t)
                oro_add_closure_tendency(g, 3, m->U[0], m->U[1], m->U[2], m->U[3 + t], m->kappa[t], m->Gn[3 + t], NULL);
        }
        if (m->has_amd) {
            for (int f = 0; f < 3; ++f) oro_add_closure_tendency_field(g, f, m->U[0], m->U[1], m->U[2], NULL, m->nu_e, m->Gn[f], NULL);
            for (int t = 0; t < m->ntr; ++t)
                oro_add_closure_tendency_field(g, 3, m->U[0], m->U[1], m->U[2], m->U[3 + t], m->kappa_e[t], m->Gn[3 + t], NULL);
        }
    }
}

/* compute_flux_bc_tendencies! (compute_nonhydrostatic_tendencies.jl:170-184): the time steppers call it right before every
 * substep (runge_kutta_3.jl:118,135,152; quasi_adams_bashforth_2.jl:99), not compute_tendencies! */
static void compute_flux_bc_tendencies(oro_model *m) {
    if (m->any_flux_bc)
        for (int f = 0; f < 3 + m->ntr; ++f) oro_compute_flux_bcs(m->g, m->Gn[f], m->loc[f], m->bcs[f]);
    /* side by side as the reference visits them: x (west, east), y, z */
    for (int f = 0; f < 3 + m->ntr; ++f)
        for (int sd = 0; sd < 6; ++sd)
            if (m->lin[f][sd].on)
                oro_compute_linear_flux_bc(m->g, m->Gn[f], m->loc[f], sd, m->lin[f][sd].a, m->lin[f][sd].b, m->U[m->lin[f][sd].dep]);
}

/* name.side = FluxBoundaryCondition((ξ, η, t, φ, p) -> a + b φ, field_dependencies = dep); -1 if the side is not Bounded, the field
 * is not at Center along it, or dep sits elsewhere in the tangential directions */
int oro_model_set_linear_flux_bc(oro_model *m, const char *name, int side, double a, double b, const char *dep) {
    char k, kd;
    const int f = field_index(m, name, &k), fd = field_index(m, dep, &kd);
    if (f < 0 || fd < 0 || k != 'U' || kd != 'U' || side < 0 || side > 5) return -1;
    const int d = side / 2;
    if (m->g->topo[d] != ORO_BOUNDED || m->loc[f][d] != ORO_CENTER) return -1;
    for (int q = 0; q < 3; ++q)
        if (q != d && m->loc[f][q] != m->loc[fd][q]) return -1;
    if (m->loc[fd][d] != ORO_CENTER) return -1;
    m->bcs[f][side].kind = ORO_BC_FLUX; m->bcs[f][side].value = 0.0;     /* halos of a Flux side: zero gradient */
    m->lin[f][side].on = 1; m->lin[f][side].dep = fd; m->lin[f][side].a = a; m->lin[f][side].b = b;
    return 0;
}

/* pressure_correction.jl:8-20 + solve_for_pressure.jl:91-95 */
static void compute_pressure_correction(oro_model *m, double dt) {
    (void)dt;
    const oro_grid *g = m->g;
    for (int f = 0; f < 3; ++f) oro_fill_halo_regions_bcs(g, m->U[f], m->loc[f], m->bcs[f], 1);
    oro_compute_source_term(g, m->U[0], m->U[1], m->U[2], oro_poisson_rhs(m->solver), m->solver->kind == 1);
    oro_poisson_solve(m->solver, m->p);
    oro_fill_halo_regions(g, m->p, LOC_C, 1);
}

/* pressure_correction.jl:40-53 */
static void make_pressure_correction(oro_model *m, double dt) {
    oro_make_pressure_correction(m->g, m->U[0], m->U[1], m->U[2], m->p);
    double dtp = fmax(2.220446049250313e-16, dt);
    oro_scale_parent(m->g, m->p, LOC_C, dtp);
}

void oro_model_set_finalize(oro_model *m, int enforce_incompressibility) {
    /* set_nonhydrostatic_model.jl:33-60: per-field fill_halo_regions! after set!, then update_state!, projection */
    for (int f = 0; f < 3 + m->ntr; ++f) oro_fill_halo_regions_bcs(m->g, m->U[f], m->loc[f], m->bcs[f], 1);
    oro_model_update_state(m, 0);
    if (enforce_incompressibility) {
        compute_pressure_correction(m, 1.0);
        make_pressure_correction(m, 1.0);
        oro_model_update_state(m, 0);
    }
}

static void tick(oro_model *m, double dt, int stage) {      /* TimeSteppers/clock.jl:128-143 */
    m->time += dt;
    if (stage) { m->stage += 1; m->last_stage_dt = dt; }
    else { m->iteration += 1; m->stage = 1; m->last_dt = dt; m->last_stage_dt = dt; }
}

static void rk3_substep(oro_model *m, double dt, double gamma, double zeta, int has_zeta) {
    for (int f = 0; f < 3 + m->ntr; ++f)
        oro_rk3_substep_field(m->g, m->U[f], m->loc[f], dt, gamma, zeta, has_zeta, m->Gn[f], m->Gm[f]);
}

static void cache_previous_tendencies(oro_model *m) {
    for (int f = 0; f < 3 + m->ntr; ++f) oro_cache_tendencies(m->g, m->Gm[f], m->Gn[f], m->loc[f]);
}

/* TimeSteppers/runge_kutta_3.jl:93-170 */
void oro_model_time_step(oro_model *m, double dt) {
    if (m->iteration == 0) oro_model_update_state(m, 1);
    const double g1 = OCN_RK3_G1, g2 = OCN_RK3_G2, g3 = OCN_RK3_G3, z2 = OCN_RK3_Z2, z3 = OCN_RK3_Z3;
    double dt1 = dt * g1, dt2 = dt * (g2 + z2), dt3 = dt * (g3 + z3);   /* :176-177 */
    double tn1 = m->time + dt;

    compute_flux_bc_tendencies(m);
    rk3_substep(m, dt, g1, 0.0, 0);
    tick(m, dt1, 1);
    compute_pressure_correction(m, dt1);
    make_pressure_correction(m, dt1);
    cache_previous_tendencies(m);
    oro_model_update_state(m, 1);

    compute_flux_bc_tendencies(m);
    rk3_substep(m, dt, g2, z2, 1);
    tick(m, dt2, 1);
    compute_pressure_correction(m, dt2);
    make_pressure_correction(m, dt2);
    cache_previous_tendencies(m);
    oro_model_update_state(m, 1);

    compute_flux_bc_tendencies(m);
    rk3_substep(m, dt, g3, z3, 1);
    double corrected = tn1 - m->time;
    tick(m, dt3, 0);
    m->last_stage_dt = corrected;
    m->last_dt = dt;
    compute_pressure_correction(m, dt3);
    make_pressure_correction(m, dt3);
    oro_model_update_state(m, 1);
}

/* TimeSteppers/quasi_adams_bashforth_2.jl:74-175: χ = 0.1 by default; forward Euler (χ = -0.5) when Δt differs
 * from clock.last_Δt (first step: last_Δt = Inf) or on request. ab2_step_field! :160-173:
 *   Gu = (1.5 + χ) Gⁿ - (0.5 + χ) G⁻ * not_euler ;  u += Δt Gu   (`* false` is a strong zero) */
void oro_ab2_step_field(const oro_grid *g, double *U, const int loc[3], double dt, double chi, const double *Gn, const double *Gm) {
    int r[6];
    default_range(g, loc, 1, r);
    fld Uf = mkfld(g, U, loc), Gnf = mkfld(g, Gn, loc), Gmf = mkfld(g, Gm, loc);
    const int not_euler = chi != -0.5;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = r[4]; k <= r[5]; ++k)
        for (int j = r[2]; j <= r[3]; ++j)
            for (int i = r[0]; i <= r[1]; ++i) {
                const double prev = not_euler ? (0.5 + chi) * AT(Gmf, i, j, k) * 1.0 : 0.0;
                const double Gu = (1.5 + chi) * AT(Gnf, i, j, k) - prev;
                AT(Uf, i, j, k) += dt * Gu;
            }
}

void oro_model_time_step_ab2(oro_model *m, double dt, double chi, int euler) {
    if (m->iteration == 0) oro_model_update_state(m, 1);
    euler = euler || (dt != m->last_dt);
    const double x = euler ? -0.5 : chi;
    compute_flux_bc_tendencies(m);
    for (int f = 0; f < 3 + m->ntr; ++f) oro_ab2_step_field(m->g, m->U[f], m->loc[f], dt, x, m->Gn[f], m->Gm[f]);
    tick(m, dt, 0);
    compute_pressure_correction(m, dt);
    make_pressure_correction(m, dt);
    cache_previous_tendencies(m);
    oro_model_update_state(m, 1);
}

/* Advection/cell_advection_timescale.jl:13-34: minimum over cells of 1 / (|u|/Δxᶠ + |v|/Δyᶠ + |w|/Δzᶠ) with Δ⁻¹ = 1/Δ; a Flat
 * direction contributes 0 */
double oro_cell_advection_timescale(const oro_grid *g, const double *u, const double *v, const double *w) {
    fld U = mkfld(g, u, LOC_U), V = mkfld(g, v, LOC_V), W = mkfld(g, w, LOC_W);
    double tau = INFINITY;
    for (int k = 1; k <= g->N[2]; ++k)
        for (int j = 1; j <= g->N[1]; ++j)
            for (int i = 1; i <= g->N[0]; ++i) {
                const double ix = g->topo[0] == ORO_FLAT ? 0.0 : fabs(AT(U, i, j, k)) * (1.0 / DF(g, 0, i));
                const double iy = g->topo[1] == ORO_FLAT ? 0.0 : fabs(AT(V, i, j, k)) * (1.0 / DF(g, 1, j));
                const double iz = g->topo[2] == ORO_FLAT ? 0.0 : fabs(AT(W, i, j, k)) * (1.0 / DF(g, 2, k));
                const double t = 1.0 / ((ix + iy) + iz);
                if (t < tau) tau = t;
            }
    return tau;
}
double oro_model_cell_advection_timescale(oro_model *m) { return oro_cell_advection_timescale(m->g, m->U[0], m->U[1], m->U[2]); }

double oro_model_time(const oro_model *m) { return m->time; }
int oro_model_iteration(const oro_model *m) { return m->iteration; }

double oro_model_max_abs_divergence(oro_model *m) {
    const oro_grid *g = m->g;
    size_t tot = (size_t)g->N[0] * g->N[1] * g->N[2];
    cplx *tmp = (cplx *)malloc(sizeof(cplx) * tot);
    oro_compute_source_term(g, m->U[0], m->U[1], m->U[2], tmp, 0);
    double mx = 0;
    for (size_t q = 0; q < tot; ++q) { double a = fabs(creal(tmp[q])); if (a > mx) mx = a; }
    free(tmp);
    return mx;
}

void oro_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int oro_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
