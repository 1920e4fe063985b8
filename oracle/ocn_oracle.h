/* ocn_oracle.h -- CPU restatement ("oracle") of the Oceananigans v0.100.5 NonhydrostaticModel RK3 time-step.
 *
 * TEST INFRASTRUCTURE ONLY. This is the checker the HIP path is compared against; nothing in the product
 * (oldoceananigans.jl_amd/, include/) may call into it. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load this library.
 *
 * PARITY STATUS: "parity unpinned" against the Julia reference for the WENO-5/RK3 floating-point fields -- there is
 * no Julia in this container or on the GPU box, and the reference's stored regression data are remote DataDeps
 * (test/data_dependencies.jl:17-38), AB2+Centered only. The restatement is pinned by the reference's data-free
 * tests instead: exact halo tests, Poisson residual tests, tridiagonal-vs-dense, incompressibility,
 * Taylor-Green, WENO order-of-accuracy/symmetry, docstring KATs; and by the numbers the reference itself holds: the stretched-
 * coordinate spacings printed with 17 digits (docs/src/fields.md), the exact ScalarDiffusivity flux divergences of
 * test/test_turbulence_closures.jl:36-66, the permutation tables, the grid summaries (tests/test_reference_kats.py,
 * tests/test_gpu_reference_tests.py). Grid generation, index work and the closure operators are therefore pinned; the WENO-5 /
 * RK3 field values are not.
 *
 * Conventions: all (i, j, k) are Julia 1-based interior indices, halo cells have indices <= 0 or > N, exactly as the
 * OffsetArrays of the reference (src/Grids/new_data.jl:15-73). Arrays are column-major (x fastest) dense parents.
 */
#ifndef OCN_ORACLE_H
#define OCN_ORACLE_H
#include <stddef.h>
#include <complex.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORO_PERIODIC = 0, ORO_BOUNDED = 1, ORO_FLAT = 3 };   /* Flat: N = 1, H = 0, Δ = 1 (Grids/grid_utils.jl); 2 is the product's FullyConnected */
enum { ORO_CENTER = 0, ORO_FACE = 1 };

typedef struct {
    int N[3], H[3], topo[3];
    double L[3];
    /* spacings with halos, index idx in 1-H .. N+H(+1): dc[d][idx - 1 + H[d]], df[d][idx - 1 + H[d]].
     * Lengths: dc N+2H+1 (last entry padding), df N+2H+1. (src/Grids/grid_generation.jl:34-135) */
    double *dc[3];
    double *df[3];
    /* buffer of the advection scheme each direction ends up with after adapt_advection_order (set by oro_grid_create
     * from WENO(order=5): 3 = WENO{3}, 2 = WENO{2}, 1 = UpwindBiased{1}) */
    int B[3];
} oro_grid;

/* Advection/adapt_advection_order.jl:62-96 for one direction: scheme family (ORO_ADV_CENTERED / _UPWIND / _WENO) with buffer B
 * on N cells of the given topology -> buffer of the adapted scheme (Flat: unchanged; N >= B: unchanged; else Centered(order=2N)
 * -> buffer N, UpwindBiased / WENO(order=2N-1) -> buffer N). required_halo_size of a scheme is its buffer (Advection.jl:63-65). */
enum { ORO_ADV_CENTERED = 0, ORO_ADV_UPWIND = 1, ORO_ADV_WENO = 2 };
int oro_adapt_advection_order(int family, int B, int N, int topo);
/* Solvers/index_permutations.jl:18-35 (1-based, as the reference writes them) */
int oro_permute_index(int i, int N);
int oro_unpermute_index(int i, int N);
/* the reference's GPU cosine transforms (discrete_transforms.jl:48-75,126-139,166-169): permute_index + FFT + twiddle (forward),
 * twiddle + IFFT + unpermute_index (backward, scaled like FFTW's REDFT01 / 2N... see the .c file); checks that route against
 * the direct REDFT10 / REDFT01 sums the rest of the oracle uses. x: n real numbers, in place. */
void oro_dct_makhoul(double *x, int n, int backward);
void oro_dct_direct(double *x, int n, int backward);

typedef struct oro_model oro_model;

/* ---- grid ---- */
oro_grid *oro_grid_create(const int N[3], const int H[3], const int topo[3], const double L[3],
                          const double *dxc, const double *dxf, const double *dyc, const double *dyf,
                          const double *dzc, const double *dzf);
void oro_grid_destroy(oro_grid *g);
/* parent extents of a field at location loc (src/Grids/grid_utils.jl:66-72) */
void oro_parent_size(const oro_grid *g, const int loc[3], int P[3]);

/* ---- halo fills (src/BoundaryConditions/fill_halo_regions*.jl) ---- */
/* default prognostic/auxiliary BCs (field_boundary_conditions.jl:15-25): Periodic -> periodic copy; Bounded+Center
 * -> no-flux mirror (one cell); Bounded+Face -> impenetrable (wall value 0), skipped when fill_open_bcs == 0 */
void oro_fill_halo_regions(const oro_grid *g, double *c, const int loc[3], int fill_open_bcs);

/* non-default boundary conditions with constant values on Bounded sides (boundary_condition_classifications.jl,
 * fill_halo_regions_value_gradient.jl:7-119, fill_halo_regions_flux.jl:9-27, fill_halo_regions_open.jl:2-7,
 * compute_flux_bcs.jl:57-163). Sides are ordered west, east, south, north, bottom, top. */
enum { ORO_BC_DEFAULT = 0, ORO_BC_FLUX = 1, ORO_BC_VALUE = 2, ORO_BC_GRADIENT = 3, ORO_BC_OPEN = 4 };
/* array != NULL: getbc(condition::AbstractArray, i, j, grid, args...) = condition[i, j] (BoundaryConditions/boundary_condition.jl:164):
 * a dense column-major array over the interior extents of the two tangential directions (x before y before z) replaces `value` */
typedef struct { int kind; double value; const double *array; } oro_bc;
void oro_fill_halo_regions_bcs(const oro_grid *g, double *c, const int loc[3], const oro_bc bcs[6], int fill_open_bcs);
/* compute_x/y/z_bcs!: adds the flux divergence of Flux boundary conditions to the tendency G of a field at loc */
void oro_compute_flux_bcs(const oro_grid *g, double *G, const int loc[3], const oro_bc bcs[6]);

/* ---- tendencies (src/Models/NonhydrostaticModels/compute_nonhydrostatic_tendencies.jl:49-163) ---- */
/* range = {i0, i1, j0, j1, k0, k1} inclusive (KernelParameters); NULL -> :xyz with exclude_periphery as the
 * reference launches it. */
void oro_compute_Gu(const oro_grid *g, const double *u, const double *v, const double *w, double *Gu, const int *range);
void oro_compute_Gv(const oro_grid *g, const double *u, const double *v, const double *w, double *Gv, const int *range);
void oro_compute_Gw(const oro_grid *g, const double *u, const double *v, const double *w, double *Gw, const int *range);
void oro_compute_Gc(const oro_grid *g, const double *u, const double *v, const double *w, const double *c, double *Gc,
                    const int *range);

/* ScalarDiffusivity(ν, κ): isotropic constant explicit closure. Adds -∂ⱼτᵢⱼ (which = 0, 1, 2: u, v, w with
 * coef = ν) or -∇·q (which = 3: tracer c with coef = κ) to a tendency that already holds the advective part */
void oro_add_closure_tendency(const oro_grid *g, int which, const double *u, const double *v, const double *w, const double *c,
                              double coef, double *G, const int *range);
/* the same with the coefficient read from a ccc array with filled halos (eddy viscosity / diffusivity) */
void oro_add_closure_tendency_field(const oro_grid *g, int which, const double *u, const double *v, const double *w, const double *c,
                                    const double *coef_ccc, double *G, const int *range);
/* AnisotropicMinimumDissipation: νₑ and κₑ[t] over the interior (anisotropic_minimum_dissipation.jl:152-216); halos are the caller's */
void oro_compute_amd_diffusivities(const oro_grid *g, double Cnu, const double *Ckappa, const double *u, const double *v,
                                   const double *w, const double *const *tracers, int ntracers, double *nu_e, double *const *kappa_e);

/* single-point WENO kernels exported for KATs */
double oro_weno5_biased(const double S[6], int left);
double oro_weno3_biased(const double S[4], int left);
double oro_newton_div_f32(double a, double b);

/* ---- RK3 (src/TimeSteppers/runge_kutta_3.jl:179-226, store_tendencies.jl:6-22) ---- */
void oro_rk3_substep_field(const oro_grid *g, double *U, const int loc[3], double dt, double gamma, double zeta,
                           int has_zeta, const double *Gn, const double *Gm);
void oro_cache_tendencies(const oro_grid *g, double *Gm, const double *Gn, const int loc[3]);

/* ---- pressure (src/Models/NonhydrostaticModels/{solve_for_pressure,pressure_correction}.jl) ---- */
void oro_compute_source_term(const oro_grid *g, const double *u, const double *v, const double *w,
                             double _Complex *rhs, int weight_by_dz);
void oro_make_pressure_correction(const oro_grid *g, double *u, double *v, double *w, const double *p);
void oro_scale_parent(const oro_grid *g, double *p, const int loc[3], double divisor);

/* ---- solvers (src/Solvers) ---- */
typedef struct oro_poisson oro_poisson;
/* kind 0: FFTBasedPoissonSolver (all dims regular); kind 1: FourierTridiagonalPoissonSolver (z tridiagonal) */
oro_poisson *oro_poisson_create(const oro_grid *g, int kind);
void oro_poisson_destroy(oro_poisson *s);
double _Complex *oro_poisson_rhs(oro_poisson *s);     /* storage (kind 0) / source_term (kind 1), size Nx*Ny*Nz */
void oro_poisson_solve(oro_poisson *s, double *phi);  /* phi: haloed (C,C,C) field, interior overwritten */
void oro_batched_tridiagonal_solve_z(int Nx, int Ny, int Nz, const double *a, const double *b3d, const double *c,
                                     const double _Complex *f, double *t, double _Complex *phi);
void oro_poisson_eigenvalues(int N, double L, int topo, double *lam);
/* unnormalised transforms on a length-n line, exported for tests: FFT (sign -1 fwd, +1 bwd), REDFT10, REDFT01 */
void oro_fft_line(double _Complex *x, int n, int stride, int sign);

/* ---- model (src/Models/NonhydrostaticModels/nonhydrostatic_model.jl, runge_kutta_3.jl:93-170) ---- */
oro_model *oro_model_create(const oro_grid *g, int ntracers);
void oro_model_destroy(oro_model *m);
double *oro_model_field(oro_model *m, const char *name); /* "u","v","w","c0".., "p", "Gu","Gv","Gw","Gc0".., "Mu".. */
void oro_model_field_loc(oro_model *m, const char *name, int loc[3]);
/* side 0..5 = west, east, south, north, bottom, top; returns 0, or -1 for an invalid combination */
int oro_model_set_bc(oro_model *m, const char *name, int side, int kind, double value);
int oro_model_set_bc_array(oro_model *m, const char *name, int side, int kind, const double *array);   /* borrowed */
/* buoyancy: kind 0 nothing; 1 BuoyancyTracer (tracer index b); 2 SeawaterBuoyancy with LinearEquationOfState
 * (tracer indices T, S; b = g (α T - β S)). The model then carries the hydrostatic pressure anomaly "pHY". */
int oro_model_set_buoyancy(oro_model *m, int kind, int b_or_T_index, int S_index, double grav, double alpha, double beta);
void oro_update_hydrostatic_pressure(const oro_grid *g, int kind, const double *b_or_T, const double *S, double grav, double alpha,
                                     double beta, double *pHY);
void oro_add_hydrostatic_pressure_gradient(const oro_grid *g, const double *pHY, double *Gu, double *Gv);
/* coriolis = FPlane(f) (Coriolis/f_plane.jl:48-52); has = 0: coriolis = nothing */
void oro_model_set_coriolis(oro_model *m, int has, double f);
void oro_add_fplane_coriolis(const oro_grid *g, double f, const double *u, const double *v, double *Gu, double *Gv);
/* closure = ScalarDiffusivity(ν = nu, κ = kappa[tracer]) ; nu = 0 and kappa = NULL/0 -> closure = nothing */
void oro_model_set_closure(oro_model *m, double nu, const double *kappa);
/* closure = AnisotropicMinimumDissipation(Cν, Cκ per tracer; Cb = nothing); fields "nu_e", "kappa_e<t>". -1 on Flat grids */
int oro_model_set_amd(oro_model *m, double Cnu, const double *Ckappa);
/* linear field-dependent Flux condition: flux = a + b dep[i, j, k_boundary] (continuous_boundary_function.jl:128-161) */
void oro_compute_linear_flux_bc(const oro_grid *g, double *G, const int loc[3], int side, double a, double b, const double *dep);
int oro_model_set_linear_flux_bc(oro_model *m, const char *name, int side, double a, double b, const char *dep);
void oro_model_update_state(oro_model *m, int compute_tendencies);
void oro_model_set_finalize(oro_model *m, int enforce_incompressibility); /* set_nonhydrostatic_model.jl:33-60 */
void oro_model_time_step(oro_model *m, double dt);
/* QuasiAdamsBashforth2TimeStepper (TimeSteppers/quasi_adams_bashforth_2.jl:74-175), χ default 0.1 */
void oro_model_time_step_ab2(oro_model *m, double dt, double chi, int euler);
void oro_ab2_step_field(const oro_grid *g, double *U, const int loc[3], double dt, double chi, const double *Gn, const double *Gm);
/* cell_advection_timescale(grid, velocities) (Advection/cell_advection_timescale.jl:13-34) */
double oro_cell_advection_timescale(const oro_grid *g, const double *u, const double *v, const double *w);
double oro_model_cell_advection_timescale(oro_model *m);
double oro_model_time(const oro_model *m);
int oro_model_iteration(const oro_model *m);
double oro_model_max_abs_divergence(oro_model *m);

void oro_set_num_threads(int n);
int oro_get_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
