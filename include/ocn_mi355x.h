/* ocn_mi355x.h -- C ABI of libocn_mi355x.so: the MI355X-native NonhydrostaticModel RK3 time-step hot path.
 *
 * This is the drop-in boundary. The reference (Oceananigans v0.100.5) has no FFI: its seam is Julia dispatch on the
 * architecture type (ext/OceananigansAMDGPUExt.jl:34-113) + `launch!(arch, grid, workspec, kernel!, args...)`
 * (src/Utils/kernel_launching.jl:340-380). A Julia maintainer binds these entry points with `ccall` from methods
 * specialised on a new architecture tag; each entry point below names the reference function
 * (file:line, relative to the reference's src/) whose body it replaces.
 *
 * Conventions
 *  - All array arguments are DEVICE pointers to dense column-major parent arrays WITH halos, exactly the memory of
 *    `parent(field.data)` (src/Grids/new_data.jl:36-73): element (i, j, k) (1-based interior index) lives at
 *    (i-1+Hx) + Px*((j-1+Hy) + Py*(k-1+Hz)), P = N + 2H (+1 for Face fields on Bounded dims, grid_utils.jl:66-72).
 *  - Pointers are BORROWED for the duration of the call; the library never frees caller memory. Objects created by
 *    *_create are owned by the library until *_destroy.
 *  - Every entry point returns int: 0 = ok; negative = invalid argument (mirrors the ArgumentErrors the reference
 *    throws at construction time); positive = hipError_t / hipfftResult (+1000) / ncclResult_t (+2000).
 *    ocn_last_error() returns a thread-local message. The library never aborts.
 *  - All work is enqueued on ONE non-blocking HIP stream per device (the reference runs every kernel on the default
 *    stream in program order, kernel_launching.jl:335-336). Entry points are asynchronous w.r.t. the host unless
 *    stated otherwise; ocn_sync() is `sync_device!` (ext/OceananigansAMDGPUExt.jl:112-113).
 *  - location codes: 0 = Center, 1 = Face. topology codes: 0 = Periodic, 1 = Bounded, 2 = FullyConnected (x), 3 = Flat.
 */
#ifndef OCN_MI355X_H
#define OCN_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCN_PERIODIC 0
#define OCN_BOUNDED 1
#define OCN_CONNECTED 2     /* FullyConnected: halo owned by a neighbouring rank (distributed_grids.jl:339-346); x only */
#define OCN_RIGHT_CONNECTED 4   /* first rank of a Bounded partitioned x: wall on the WEST side, neighbour on the east (distributed_grids.jl:339-346) */
#define OCN_LEFT_CONNECTED 5    /* last rank: neighbour on the west side, wall on the EAST side; Face-in-x fields hold Nx + 1 faces (grid_utils.jl:43-68) */
#define OCN_FLAT 3          /* Flat: size 1, halo 0, unit spacing and extent; differences 0, interpolations the identity
                             * (Grids/grid_utils.jl, Operators/difference_operators.jl:30-49, Advection/flat_advective_fluxes.jl) */
#define OCN_CENTER 0
#define OCN_FACE 1

#define OCN_OK 0
#define OCN_EINVAL (-1)      /* invalid argument (ArgumentError in the reference) */
#define OCN_ENOTSUP (-2)     /* configuration outside the accelerated hot path */
#define OCN_ESTATE (-3)      /* call sequence error (e.g. library not initialised) */
#define OCN_EFFT (-4)        /* an FFT plan failed its creation-time round-trip self-check (rocFFT note) */

typedef struct ocn_grid_s *ocn_grid_t;
typedef struct ocn_poisson_s *ocn_poisson_t;
typedef struct ocn_model_s *ocn_model_t;
typedef struct ocn_dist_s *ocn_dist_t;
typedef struct ocn_transposable_s *ocn_transposable_t;   /* TransposableField (transposable_field.jl:5-13) */

/* ---------------------------------------------------------------- runtime (src/Architectures.jl:35-123) ---------- */
int ocn_init(int device_id);                                  /* device!(arch, id) */
int ocn_device_count(int *count);                             /* ndevices(arch) (Architectures.jl; distributed_architectures.jl:284-288 assigns node_rank % ndevices) */
int ocn_sync(void);                                           /* sync_device! */
const char *ocn_last_error(void);
const char *ocn_version(void);
int ocn_malloc(void **ptr, size_t bytes);                     /* zeros(arch, FT, dims...) -- memory is zero-filled */
int ocn_free(void *ptr);                                      /* unsafe_free! */
int ocn_memcpy_h2d(void *dst, const void *src, size_t bytes); /* on_architecture(GPU(), a)  (synchronous) */
int ocn_memcpy_d2h(void *dst, const void *src, size_t bytes); /* on_architecture(CPU(), a)  (synchronous) */
int ocn_memcpy_d2d(void *dst, const void *src, size_t bytes); /* device_copy_to! (stream ordered) */
int ocn_memset_zero(void *dst, size_t bytes);
void *ocn_stream(void);                                       /* the hipStream_t all work is enqueued on */
/* enqueue all subsequent work on a caller-owned stream (e.g. torch.cuda.current_stream(), so RCCL collectives issued
 * through torch.distributed are stream-ordered with the kernels; the reference instead calls sync_device! before every
 * MPI call, halo_communication.jl:181) */
int ocn_set_stream(void *hip_stream);
/* back to a library-owned stream (the state after ocn_init). Time-step graphs (ocn_model_time_step) need it: a borrowed stream
 * may be the legacy default stream, which cannot be captured. */
int ocn_own_stream(void);

/* ---------------------------------------------------------------- grid (src/Grids/rectilinear_grid.jl:3-25) ------ */
/* N, H, topo, L: per dimension. dx, dy: regular spacings. dzc / dzf: HOST arrays of Δzᵃᵃᶜ / Δzᵃᵃᶠ for index
 * k = 1-Hz .. Nz+Hz+1 (length Nz + 2Hz + 1, position k-1+Hz); pass NULL for a z-regular grid (then dz is used).
 * Stretched x / y are outside the hot path -> callers must pass regular dx, dy. */
int ocn_grid_create(ocn_grid_t *grid, const int N[3], const int H[3], const int topo[3], const double L[3],
                    double dx, double dy, double dz, const double *dzc, const double *dzf);
int ocn_grid_destroy(ocn_grid_t grid);
int ocn_grid_parent_size(ocn_grid_t grid, const int loc[3], int P[3]);    /* total_size, grid_utils.jl:138-169 */

/* ---------------------------------------------------------------- halo fills (BoundaryConditions/) -------------- */
/* fill_halo_regions!(field) with the default boundary conditions of field_boundary_conditions.jl:15-25
 * (Periodic -> PeriodicBC copies, fill_halo_regions_periodic.jl:5-33; Bounded+Center -> no-flux one-cell mirror,
 * fill_halo_regions_flux.jl:9-27; Bounded+Face -> impenetrable wall value, fill_halo_regions_open.jl:2-7, skipped when
 * fill_open_bcs == 0). nfields fields of identical location are filled by ONE launch. */
int ocn_fill_halo_regions(ocn_grid_t grid, double *const *fields, const int (*locs)[3], int nfields, int fill_open_bcs);

/* Non-default boundary conditions with constant values on Bounded sides (BoundaryCondition{<:Flux|Value|Gradient|Open},
 * boundary_condition.jl / boundary_condition_classifications.jl). Sides are ordered west, east, south, north, bottom, top.
 * Value / Gradient fill ONE halo cell by linear extrapolation through the boundary face
 * (fill_halo_regions_value_gradient.jl:7-119), Flux fills like no-flux (fill_halo_regions_flux.jl:9-27) and contributes
 * through ocn_compute_flux_bcs, Open sets the wall-normal component on the boundary face (fill_halo_regions_open.jl:2-7).
 * OCN_BC_DEFAULT = what field_boundary_conditions.jl:15-25 assigns. bcs[f][side]; bcs == NULL: all default. */
#define OCN_BC_DEFAULT 0
#define OCN_BC_FLUX 1
#define OCN_BC_VALUE 2
#define OCN_BC_GRADIENT 3
#define OCN_BC_OPEN 4
/* `array` != NULL: an array-valued condition, getbc(condition::AbstractArray, i, j, grid) = condition[i, j] (boundary_condition.jl:164):
 * a BORROWED device pointer to a dense column-major array over the interior extents of the two tangential directions in the order
 * x before y before z -- west / east: (Ny, Nz), south / north: (Nx, Nz), bottom / top: (Nx, Ny) -- that replaces the number `value`
 * point by point (Flux, Value, Gradient and Open conditions alike). It must stay valid while the condition is in use. */
typedef struct { int kind; double value; const double *array; } ocn_bc_t;
int ocn_fill_halo_regions_bcs(ocn_grid_t grid, double *const *fields, const int (*locs)[3], int nfields,
                              const ocn_bc_t (*bcs)[6], int fill_open_bcs);
/* compute_x_bcs! / compute_y_bcs! / compute_z_bcs! (BoundaryConditions/compute_flux_bcs.jl:12-163), called by
 * compute_flux_bc_tendencies! (compute_nonhydrostatic_tendencies.jl:170-184): G[1] += flux A / V, G[N] -= flux A / V */
int ocn_compute_flux_bcs(ocn_grid_t grid, double *G, const int loc[3], const ocn_bc_t bcs[6]);
/* One side (0 west .. 5 top) of a field-dependent Flux condition of the linear family, flux = a + b dep[i, j, k_boundary]: what
 * getbc(::ContinuousBoundaryFunction) (BoundaryConditions/continuous_boundary_function.jl:128-161) evaluates for
 * func(x, y, t, φ, p) = a + b φ with field_dependencies = :φ when φ sits at the location `loc` of the field that carries the condition
 * (identity interpolation). examples/ocean_wind_mixing_and_convection.jl:125-136: a = 0, b = -evaporation_rate, φ = S. General
 * callables cannot cross a C ABI; this family covers relaxation / evaporation / linear-drag conditions. */
int ocn_compute_linear_flux_bc(ocn_grid_t grid, double *G, const int loc[3], int side, double a, double b, const double *dep);

/* ---------------------------------------------------------------- tendencies ------------------------------------ */
/* compute_Gu!/Gv!/Gw!/Gc! (Models/NonhydrostaticModels/compute_nonhydrostatic_tendencies.jl:138-163) for
 * advection = WENO(order=5), every other term `nothing`. range = {i0,i1,j0,j1,k0,k1} inclusive 1-based
 * (KernelParameters, kernel_launching.jl:25-95) or NULL for the reference's default launch (:xyz, exclude_periphery
 * for velocities). */
int ocn_compute_Gu(ocn_grid_t grid, const double *u, const double *v, const double *w, double *Gu, const int *range);
int ocn_compute_Gv(ocn_grid_t grid, const double *u, const double *v, const double *w, double *Gv, const int *range);
int ocn_compute_Gw(ocn_grid_t grid, const double *u, const double *v, const double *w, double *Gw, const int *range);
int ocn_compute_Gc(ocn_grid_t grid, const double *u, const double *v, const double *w, const double *c, double *Gc,
                   const int *range);
/* compute_interior_tendency_contributions! (…tendencies.jl:49-131): all of Gu, Gv, Gw and ntracers Gc in one fused,
 * flux-sharing pass. */
int ocn_compute_tendencies(ocn_grid_t grid, const double *u, const double *v, const double *w,
                           const double *const *tracers, int ntracers,
                           double *Gu, double *Gv, double *Gw, double *const *Gc, const int *range);
/* the same pass with the rk3_substep! of the NEXT stage (runge_kutta_3.jl:179-226) fused in: when a cell's tendency is
 * complete, U_next = U + dt (gamma Gn + zeta Gm) is written to a SECOND set of prognostic arrays (U itself is still read by
 * neighbouring workgroups); the caller then swaps the two sets. fields / next / Gn / Gm are ordered u, v, w, tracers
 * (3 + ntracers). OCN_ENOTSUP when the fused kernel cannot run on this grid (Bounded x / y). */
int ocn_compute_tendencies_and_substep(ocn_grid_t grid, const double *const *fields, int ntracers, double *const *Gn,
                                       const int *range, double *const *next, const double *const *Gm, double dt,
                                       double gamma, double zeta, int has_zeta);

/* buoyancy, gravity along -z: kind 1 = BuoyancyTracer (bT = the buoyancy tracer), kind 2 = SeawaterBuoyancy with
 * a LinearEquationOfState, b = g (α T - β S) (bT = T). _update_hydrostatic_pressure! (update_hydrostatic_pressure.jl:12-22, over
 * i = 0:Nx+1, j = 0:Ny+1) and the -∂x pHY′, -∂y pHY′ terms of the u, v tendencies (nonhydrostatic_tendency_kernel_functions.jl:
 * 14-19,97,159), added to tendencies that hold the advective part (range: the cells whose tendencies are updated, NULL = all). */
int ocn_update_hydrostatic_pressure(ocn_grid_t grid, int kind, const double *bT, const double *S, double g, double alpha, double beta,
                                    double *pHY);
int ocn_add_hydrostatic_pressure_gradient(ocn_grid_t grid, const double *pHY, double *Gu, double *Gv, const int *range);

/* coriolis = FPlane(f) (Coriolis/f_plane.jl:48-52): G_u -= x_f_cross_U = -f active_weighted_ℑxyᶠᶜᶜ(v), G_v -=
 * y_f_cross_U = f active_weighted_ℑxyᶜᶠᶜ(u) (Operators/interpolation_operators.jl:116-130), on tendencies holding the advective part */
int ocn_add_fplane_coriolis(ocn_grid_t grid, double f, const double *u, const double *v, double *Gu, double *Gv, const int *range);

/* closure = ScalarDiffusivity(ν, κ): isotropic, constant, explicit (SURVEY.md 8f.1 -- the first "next" row).
 * ∂ⱼ_τ₁ⱼ / ∂ⱼ_τ₂ⱼ / ∂ⱼ_τ₃ⱼ / ∇_dot_qᶜ (TurbulenceClosures/closure_kernel_operators.jl:22-48) with viscous_flux_* = -2 ν Σᵢⱼ and
 * diffusive_flux_* = -κ ∂c (abstract_scalar_diffusivity_closure.jl:194-242). ADDS the closure term to tendencies that already
 * hold the advective part, in the order of nonhydrostatic_tendency_kernel_functions.jl:91-100: G = (G - ∂ⱼτᵢⱼ) + 0.
 * kappa: one value per tracer. range as in ocn_compute_tendencies. */
int ocn_compute_closure_tendencies(ocn_grid_t grid, const double *u, const double *v, const double *w,
                                   const double *const *tracers, int ntracers, double nu, const double *kappa,
                                   double *Gu, double *Gv, double *Gw, double *const *Gc, const int *range);
/* the same with the coefficients read from ccc arrays with filled halos -- the eddy viscosity / diffusivities of an LES closure,
 * interpolated to the flux locations (abstract_scalar_diffusivity_closure.jl:310-330: ν[i,j,k], ℑxyᶠᶠᵃ, ℑxzᶠᵃᶠ, ℑyzᵃᶠᶠ, ℑxᶠᵃᵃ, ...) */
int ocn_compute_closure_tendencies_field(ocn_grid_t grid, const double *u, const double *v, const double *w,
                                         const double *const *tracers, int ntracers, const double *nu_e,
                                         const double *const *kappa_e, double *Gu, double *Gv, double *Gw, double *const *Gc,
                                         const int *range);
/* compute_diffusivities!(diffusivity_fields, closure::AnisotropicMinimumDissipation, model; parameters = :xyz)
 * (turbulence_closure_implementations/anisotropic_minimum_dissipation.jl:152-216; Cb = nothing): νₑ and κₑ[t] over the interior from
 * fields with filled halos; the caller fills the halos of the results (ocn_fill_halo_regions, default conditions).
 * range: NULL = the interior (:xyz), or {i0, i1, j0, j1, k0, k1} reaching at most H - 1 cells into the halos -- an x-slab rank
 * computes i = 0 and Nx + 1 from its exchanged velocity / tracer halos instead of exchanging νₑ, κₑ. */
int ocn_compute_amd_diffusivities(ocn_grid_t grid, double Cnu, const double *Ckappa, const double *u, const double *v,
                                  const double *w, const double *const *tracers, int ntracers, double *nu_e,
                                  double *const *kappa_e, const int *range);

/* ---------------------------------------------------------------- RK3 (TimeSteppers/runge_kutta_3.jl) ----------- */
/* rk3_substep_field! (:212-226), launched with exclude_periphery (:187). has_zeta == 0 selects the first-stage
 * method `U += Δt γ¹ G¹`. */
/* ab2_step_field! (TimeSteppers/quasi_adams_bashforth_2.jl:160-173), launched with exclude_periphery: U += Δt ((3/2 + χ) Gⁿ -
 * (1/2 + χ) G⁻); χ = -0.5 is the forward-Euler step (G⁻ is not read) */
int ocn_ab2_step(ocn_grid_t grid, double *const *U, const double *const *Gn, const double *const *Gm, const int (*locs)[3],
                 int nfields, double dt, double chi);
int ocn_rk3_substep(ocn_grid_t grid, double *const *U, const double *const *Gn, const double *const *Gm,
                    const int (*locs)[3], int nfields, double dt, double gamma, double zeta, int has_zeta);
/* _cache_field_tendencies! (TimeSteppers/store_tendencies.jl:6-9) */
int ocn_cache_tendencies(ocn_grid_t grid, double *const *Gm, const double *const *Gn, const int (*locs)[3], int nfields);

/* ---------------------------------------------------------------- pressure -------------------------------------- */
/* _compute_source_term! / _fourier_tridiagonal_source_term!(ZDirection) (solve_for_pressure.jl:12-18, 36-42).
 * rhs: interleaved complex double, dense (Nx, Ny, Nz). */
int ocn_compute_source_term(ocn_grid_t grid, const double *u, const double *v, const double *w, double *rhs_complex,
                            int weight_by_dz);
/* _make_pressure_correction! (pressure_correction.jl:31-37) */
int ocn_make_pressure_correction(ocn_grid_t grid, double *u, double *v, double *w, const double *p);
/* the same over {i0, i1, j0, j1, k0, k1} (NULL = everything): an x-slab rank corrects its two boundary strips first, starts the
 * halo exchange of the next update_state! and corrects the interior while the halos are in flight */
int ocn_make_pressure_correction_range(ocn_grid_t grid, double *u, double *v, double *w, const double *p, const int *range);
/* _make_pressure_correction! and `pNHS ./= Δt⁺` (pressure_correction.jl:31-50) in one pass over `range` (NULL = everything): p / divisor
 * goes to a SECOND haloed array `p_divided` (other threads still read p), which the caller makes the pressure field afterwards */
int ocn_make_pressure_correction_divide(ocn_grid_t grid, double *u, double *v, double *w, const double *p, double *p_divided,
                                        double divisor, const int *range);
/* `pNHS ./= Δt⁺` (pressure_correction.jl:48-50): interior of a (Center, Center, Center) field */
int ocn_divide_interior(ocn_grid_t grid, double *p, double divisor);

/* ---------------------------------------------------------------- solvers (src/Solvers) ------------------------- */
/* kind 0: FFTBasedPoissonSolver (fft_based_poisson_solver.jl:52-74); kind 1: FourierTridiagonalPoissonSolver with
 * tridiagonal direction z (fourier_tridiagonal_poisson_solver.jl:75-134); kind -1: pick like
 * NonhydrostaticModels.jl:25-40 (regular z -> 0, stretched z -> 1). */
int ocn_poisson_create(ocn_poisson_t *solver, ocn_grid_t grid, int kind);
int ocn_poisson_destroy(ocn_poisson_t solver);
int ocn_poisson_kind(ocn_poisson_t solver);
/* device pointer to the complex right-hand-side storage (solver.storage / solver.source_term) */
int ocn_poisson_rhs(ocn_poisson_t solver, double **rhs_complex);
/* solve!(ϕ, solver) (fft_based_poisson_solver.jl:95-125 / fourier_tridiagonal_poisson_solver.jl:212-239): consumes the
 * rhs storage, writes the interior of the haloed (C,C,C) field phi. */
int ocn_poisson_solve(ocn_poisson_t solver, double *phi);
/* solve_for_pressure! (solve_for_pressure.jl:91-95) = source term + solve */
int ocn_solve_for_pressure(ocn_poisson_t solver, const double *u, const double *v, const double *w, double *p);
/* solve_batched_tridiagonal_system_kernel! (batched_tridiagonal_solver.jl:213-245), z direction: a, c length Nz-1,
 * b dense (Nx,Ny,Nz) real, f / phi dense complex, t dense real scratch. */
int ocn_batched_tridiagonal_solve_z(int Nx, int Ny, int Nz, const double *a, const double *b, const double *c,
                                    const double *f_complex, double *t, double *phi_complex);

/* ---------------------------------------------------------------- distributed x-slab pieces --------------------- */
/* fill_send_buffers! / recv_from_buffers! (DistributedComputations/communication_buffers.jl:281-313) for Partition(R):
 * west/east buffers of Hx x Py x Pz doubles per field (whole parent extent in y, z: corners ride along; Py, Pz are
 * each field's own -- Face fields on Bounded dimensions have one more plane), field-major, back to back.
 * The exchange itself (MPI.Isend/Irecv in the reference, halo_communication.jl:300,326) is issued by the host layer
 * over RCCL. */
int ocn_pack_x_halos(ocn_grid_t grid, double *const *fields, const int (*locs)[3], int nfields, double *west_send,
                     double *east_send);
int ocn_unpack_x_halos(ocn_grid_t grid, double *const *fields, const int (*locs)[3], int nfields, const double *west_recv,
                       const double *east_recv);
/* the same with only the `depth` (1 <= depth <= Hx) columns next to each side: buffers of depth x Py x Pz doubles per field.
 * The pressure solve reads one column of x halo (u[Nx+1] in the divergence, p[0] in the correction): the host layer exchanges
 * exactly that between the full fills of update_state!, where the reference's generic fill moves Hx columns of u, v, w
 * (pressure_correction.jl:8-20) that nothing reads before they are filled again. */
int ocn_pack_x_halos_depth(ocn_grid_t grid, double *const *fields, const int (*locs)[3], int nfields, int depth,
                           double *west_send, double *east_send);
int ocn_unpack_x_halos_depth(ocn_grid_t grid, double *const *fields, const int (*locs)[3], int nfields, int depth,
                             const double *west_recv, const double *east_recv);
/* DistributedFFTBasedPoissonSolver (distributed_fft_based_poisson_solver.jl:92-188; z Periodic) and
 * DistributedFourierTridiagonalPoissonSolver (distributed_fft_tridiagonal_solver.jl:153-293; z Bounded, regular or
 * stretched) for Partition(R,1,1), split at the two transposes (MPI.Alltoallv!, distributed_transpose.jl:185-191) which
 * the host layer runs as RCCL all-to-alls on the send/recv buffers:
 *   source_term -> forward_yz -> [all_to_all(recv, send)] -> solve_x -> [all_to_all(recv, send)] -> backward_yz -> phi.
 * The right-hand side is real, so only the y modes 0..Ny/2 travel (half the reference's bytes): send/recv hold
 * `buffer_size` complex elements = R equal chunks of (Nxl, ceil((Ny/2+1)/R), Nz). */
typedef struct ocn_dist_poisson_s *ocn_dist_poisson_t;
int ocn_dist_poisson_create(ocn_dist_poisson_t *solver, ocn_grid_t local_grid, int R, int rank, double Lx_global);
int ocn_dist_poisson_destroy(ocn_dist_poisson_t solver);
int ocn_dist_poisson_buffer_size(ocn_dist_poisson_t solver, size_t *complex_elements);
int ocn_dist_poisson_set_buffers(ocn_dist_poisson_t solver, double *send_complex, double *recv_complex);
/* compute_source_term! (solve_for_pressure.jl:12-84) into the solver's own storage */
int ocn_dist_poisson_source_term(ocn_dist_poisson_t solver, const double *u, const double *v, const double *w);
int ocn_dist_poisson_forward_yz(ocn_dist_poisson_t solver);
/* z Periodic and option "dist_substructured" = 1 (default): after the local (y, z) transform every mode is a periodic constant-
 * coefficient tridiagonal system along the partitioned x direction; it is solved by substructuring -- Thomas sweeps on the slab, an
 * ALL-GATHER of 2 values per mode (payload_size complex elements per rank, ~1 MB at 256^3) instead of the two all-to-alls of the
 * whole spectrum, a 2x2 solve per mode and rank-DFT index, a slab correction:
 *   source_term -> forward_local -> [all_gather(gathered, payload)] -> backward_local -> phi.
 * Same solution as the transposed FFT solve to round-off. payload_size = 0: the solver transposes (stages above). */
int ocn_dist_poisson_payload_size(ocn_dist_poisson_t solver, size_t *complex_elements);
/* which local layout the substructured solve runs on (diagnostic): 3 = z-fastest real array, 1-D R2C / C2R plans along z and the y
 * pass by the library's own LDS column-FFT kernel (Ny = 2^m <= 512; default); 2 = z-fastest real array, ONE 2-D (y, z) R2C / C2R plan batched over
 * the local x index, spectrum already in the order of the Thomas sweeps; 1 = the same with 1-D plans (rocFFT refuses the
 * 2-D interleaved-batch layout for some small sizes); 0 = paired real columns + Hermitian separation (option dist_zfirst = 0);
 * -1 = transposing solver */
int ocn_dist_poisson_layout(ocn_dist_poisson_t solver, int *layout);
int ocn_dist_poisson_set_gather_buffers(ocn_dist_poisson_t solver, double *payload_complex, double *gathered_complex);
int ocn_dist_poisson_forward_local(ocn_dist_poisson_t solver);
int ocn_dist_poisson_backward_local(ocn_dist_poisson_t solver, double *phi);
int ocn_dist_poisson_solve_x(ocn_dist_poisson_t solver);
int ocn_dist_poisson_backward_yz(ocn_dist_poisson_t solver, double *phi);

/* ---------------------------------------------------------------- model fast path ------------------------------- */
/* NonhydrostaticModel(; grid, advection = WENO(), tracers, timestepper = :RungeKutta3) with coriolis / buoyancy /
 * closure / forcing = nothing (nonhydrostatic_model.jl:115-244). Fields are allocated (zeroed) by the library. */
int ocn_model_create(ocn_model_t *model, ocn_grid_t grid, int ntracers);
int ocn_model_destroy(ocn_model_t model);
/* names: "u","v","w","c0".."c7" (tracers), "p" (pNHS), "Gu".."Gc7" (Gⁿ), "Mu".."Mc7" (G⁻). Returns the device
 * pointer of the parent array and its location. Pointers stay valid for the model's lifetime and are stable at
 * time-step boundaries. */
int ocn_model_field(ocn_model_t model, const char *name, double **ptr, int loc[3]);
/* update_state!(model; compute_tendencies) (update_nonhydrostatic_model_state.jl:20-56). As in the reference the tendencies it leaves carry
 * no Flux-boundary-condition terms: compute_flux_bc_tendencies! belongs to the stage that follows (runge_kutta_3.jl:118,134,150). */
int ocn_model_update_state(ocn_model_t model, int compute_tendencies);
/* tail of set!(model; ...) after the interiors were written (set_nonhydrostatic_model.jl:44-57) */
int ocn_model_set_finalize(ocn_model_t model, int enforce_incompressibility);
/* time_step!(model, Δt) (runge_kutta_3.jl:93-170). One call = the three stages; what it leaves is what the reference leaves: the fields,
 * pNHS of the third stage, Gⁿ = G(U³) without Flux-condition terms, G⁻ = G(U¹). Intermediate values nothing can read from outside the call
 * (pNHS of stages 1 and 2, the tendency of the second stage) are not stored: options "skip_stage_pressure", "skip_dead_tendency_store". */
int ocn_model_time_step(ocn_model_t model, double dt);
/* time_step!(model::AbstractModel{<:QuasiAdamsBashforth2TimeStepper}, Δt; euler) (TimeSteppers/quasi_adams_bashforth_2.jl:74-123;
 * SURVEY.md 8f.1): χ = 0.1 is the reference's default; a forward-Euler step is taken when Δt differs from clock.last_Δt (first
 * step) or euler != 0 */
int ocn_model_time_step_ab2(ocn_model_t model, double dt, double chi, int euler);
/* reset!(model.clock); reset!(model.timestepper) (Simulations/simulation.jl:203-213): time = 0, iteration = 0, stage = 1,
 * last_Δt = Inf, Gⁿ = G⁻ = 0. The next time-step starts with update_state! again. */
int ocn_model_reset(ocn_model_t model);
int ocn_model_clock(ocn_model_t model, double *time, int64_t *iteration, int *stage, double *last_dt,
                    double *last_stage_dt);
/* set!(model, checkpointed_clock) (OutputWriters/checkpointer.jl:199-231): restore the clock of a checkpointed state; the fields and
 * the tendencies Gⁿ, G⁻ are restored by copying the checkpointed parent arrays (halos included, like the reference's files) into the
 * arrays ocn_model_field returns, followed by ocn_model_update_state */
int ocn_model_set_clock(ocn_model_t model, double time, int64_t iteration, int stage, double last_dt, double last_stage_dt);
/* max |∇·u| over the interior (test helper: test/test_time_stepping.jl:124-160); synchronous */
int ocn_model_max_abs_divergence(ocn_model_t model, double *value);
/* cell_advection_timescale(grid, velocities) (Advection/cell_advection_timescale.jl:13-34): min over cells of
 * 1 / (|u|/Δx + |v|/Δy + |w|/Δz) -- what TimeStepWizard multiplies by the CFL number. Synchronous. */
int ocn_cell_advection_timescale(ocn_grid_t grid, const double *u, const double *v, const double *w, double *tau);
int ocn_model_cell_advection_timescale(ocn_model_t model, double *tau);
/* hasnan(field) = any(isnan, parent(field)) (Diagnostics/nan_checker.jl:32): `n` doubles starting at `data` (the whole parent
 * array, halos included); *result = 1 if any is NaN */
int ocn_hasnan(const double *data, size_t n, int *result);
int ocn_max_abs_divergence(ocn_grid_t grid, const double *u, const double *v, const double *w, double *value);
/* options: "tendency_impl" 0 = per-field kernels as the reference launches them, 1 = fused flux-sharing kernel;
 * "swap_tendencies" 1 = cache_previous_tendencies! by pointer swap, 0 = by copy kernel; "fuse_substep" 1 = fuse the
 * substeps of RK3 stages 2 and 3 into the preceding tendency evaluation (second set of prognostic arrays, swapped twice per
 * time-step); "profile" 1 = record HIP
 * events around every tendency evaluation on the launch stream */
int ocn_model_set_option(ocn_model_t model, const char *key, int value);
/* reads an option back; additionally "fused_tendency_active" (1 when the fused flux-sharing kernel runs for this grid) and
 * "fuse_substep_active" (1 when the rk3_substep! of stages 2 and 3 is fused into the preceding tendency evaluation:
 * option "fuse_substep" = 1 (default), fused kernel, tendencies cached by pointer swap, no Flux boundary condition) */
int ocn_model_get_option(ocn_model_t model, const char *key, int *value);
/* boundary_conditions = (name = FieldBoundaryConditions(side = BoundaryCondition(kind, value)),) of the model
 * constructor (nonhydrostatic_model.jl:115-244); name "u","v","w","c0".. and, with an LES closure, the diffusivity fields "nu_e",
 * "kappa_e0".. (Value / Gradient; boundary_conditions = (κₑ = (b = ...,),)); side 0..5 = west .. top. OCN_EINVAL mirrors
 * the reference's validation: Bounded sides only; Flux/Value/Gradient on Center-located, Open on Face-located fields */
/* buoyancy = nothing (kind 0) | BuoyancyTracer() (kind 1, tracer index) | SeawaterBuoyancy(LinearEquationOfState(α, β), g)
 * (kind 2, tracer indices of T and S). With buoyancy the model carries the hydrostatic pressure anomaly, field name "pHY"
 * (nonhydrostatic_model.jl:144-158). */
int ocn_model_set_buoyancy(ocn_model_t model, int kind, int b_or_T_index, int S_index, double g, double alpha, double beta);
/* coriolis = FPlane(f = f) of the model constructor; enabled = 0: coriolis = nothing */
int ocn_model_set_coriolis(ocn_model_t model, int enabled, double f);
/* closure = ScalarDiffusivity(ν = nu, κ = kappa[tracer]) of the model constructor; all zeros / NULL: closure = nothing */
int ocn_model_set_closure(ocn_model_t model, double nu, const double *kappa);
/* closure = AnisotropicMinimumDissipation(Cν = Cnu, Cκ = Ckappa[tracer]) (replaces a ScalarDiffusivity). update_state! then
 * computes the model fields "nu_e", "kappa_e0", ... and fills their halos before the tendencies. */
int ocn_model_set_amd(ocn_model_t model, double Cnu, const double *Ckappa);
int ocn_model_set_boundary_condition(ocn_model_t model, const char *name, int side, int kind, double value);
/* the same with an array-valued condition (see ocn_bc_t; borrowed device pointer, valid for the model's lifetime) */
int ocn_model_set_boundary_condition_array(ocn_model_t model, const char *name, int side, int kind, const double *device_array);
/* name.side = FluxBoundaryCondition((ξ, η, t, φ, p) -> a + b φ, field_dependencies = dep); dep at the location of `name` */
int ocn_model_set_linear_flux_bc(ocn_model_t model, const char *name, int side, double a, double b, const char *dep);
/* library-wide knobs (no reference equivalent; the defaults are the tuned values, 0 / 1 unless noted):
 *   pressure solve: "real_fft" (1: D2Z/Z2D, 0: the reference's complex-to-complex), "c2r_strided", "fused_zfft" (z FFT + divide +
 *     inverse z FFT as one LDS pass), "split_solve" (model time-step: 1-D x plans on 128-B-padded rows + LDS column-FFT kernel for y +
 *     pressure correction and p/Δt from the dense solution), "skip_stage_pressure" (1: RK3 stages 1 and 2 do not store their pNHS --
 *     nothing can read it before the next stage overwrites it; after a time-step the field holds the last stage's pressure either way),
 *     "skip_dead_tendency_store" (1: the tendency evaluated after RK3's second stage feeds the third stage's substep riding along and is
 *     not stored -- nothing else reads it; after a time-step Gⁿ = G(U³) and G⁻ = G(U¹) either way);
 *   x-slab solve: "dist_substructured" (1: gathered interface solve, 0: the reference's two transposes), "dist_zfirst" (z-fastest
 *     local layout), "dist_yline" (LDS column-FFT kernel for the local y transform), "dist_xfast" (the substructured solve in the fields'
 *     own x-fastest layout), "dist_fuse_source" (source term written straight into the z transform's line buffer), "dist_xline_group"
 *     (several short x lines per wave in the Thomas scans), "dist_fused_step" (pressure step without fills / copies between its stages),
 *     "dist_pencil_transposes" (pencil partitions: the reference's transposing solver; 0: gathered solve), "line_zl512" (4 | 8 lines per
 *     workgroup of the LDS line transforms at 512-point lines);
 *   physics passes: "amd_march" / "epilogue_march" (z-marching eddy-diffusivity kernel / tendency epilogue that evaluate every point
 *     operand / face flux once; 0: one thread per value, everything recomputed -- same bits), "epilogue_kchunk" (levels per workgroup,
 *     0 = automatic), "epilogue_rows" (rows per workgroup, 1 .. 8);
 *   fused tendency kernel: "fused_ty" (tile rows 3 | 7), "fused_kchunk" (levels per workgroup, 0 = automatic), "fused_minw",
 *     "fused_zwin" (register z-windows), "fused_xcd" (XCD-aware tile order; measured: no effect);
 *   halo fills: "fused_halo" (one launch per periodic fill);
 *   "arithmetic": 0 (default) the reference's IEEE operation sequence in every kernel -- results bit-identical to a faithful CPU
 *     evaluation of weno_interpolants.jl; 1 the opt-in CONTRACTED WENO-5 flux of the flux-sharing tendency kernel (fma-contracted
 *     sub-stencil polynomials and alpha weights, one normalisation of the weighted sum, reciprocal without the IEEE divide's fix-up,
 *     advecting transport multiplied by the area after its interpolation): fewer FP64 instructions, fields within 1e-12 of mode 0
 *     on O(1) data but not bit-identical.
 * Model options (ocn_model_set_option): "tendency_impl" (1 fused, 0 per-field kernels), "swap_tendencies", "fuse_substep",
 * "fused_epilogue", "use_graph" (hipGraph replay of the RK3 step; measured: no gain, default 0), "profile". */
int ocn_set_option(const char *key, int value);
/* sum of the event-timed tendency evaluations since the last read (ms) and their count; synchronous. This is the
 * measurement hook the reference lacks (it is profiled externally with nsys, .buildkite/pipeline-benchmarks.yml:57) */
int ocn_model_profile_read(ocn_model_t model, double *tendency_ms, int *count);

/* test hook: counts the Float32 significands (of 2^23, binade 2^exponent) for which the fast correctly-rounded
 * reciprocal used inside newton_div (Utils/newton_div.jl:8-20) differs from the IEEE divide; must return 0 */
int ocn_debug_rcp_check(int variant, int exponent, unsigned long long *mismatches);
/* sampled check of the Float64 reciprocal used for the WENO weight normalisation against the compiler's IEEE divide:
 * nsamples pseudo-random significands with exponents exp_lo..exp_hi; returns the number of differing results */
int ocn_debug_rcp64_check(unsigned long long nsamples, int exp_lo, int exp_hi, unsigned long long seed,
                          unsigned long long *mismatches);
/* where the cosine-transform path's permute_indices! (backward = 0) / unpermute_indices! (backward = 1) send element i of a line of
 * length N: destination[i - 1], 1-based -- the tables of Solvers/index_permutations.jl:5-35 (host array, synchronous) */
int ocn_debug_permute_indices(int N, int backward, int *destination);
/* number of Poisson solvers of this process that switched to the per-direction transform path because their multi-dimensional rocFFT
 * plans failed the creation-time self-check (rocFFT returns wrong transforms from such plans while plans of other sizes are alive;
 * the unit-stride 1-D plans of the per-direction path are not affected) */
int ocn_debug_fft_fallbacks(void);

/* ---------------------------------------------------------------- distributed: communicator + partitioned model -- */
/* `Distributed(GPU(); partition = Partition(R))` (DistributedComputations/distributed_architectures.jl:166-302): one process per
 * GPU, x-slabs, ring neighbours with periodic wrap (:391-434). The library owns the communicator: RCCL over xGMI
 * (ncclCommInitRank; librccl is opened at run time by the first of these calls, single-GPU users never load it). Bootstrap as
 * with NCCL: rank 0 calls ocn_dist_unique_id, the caller carries the 128 bytes to every rank (MPI_Bcast, a TCP store, a file),
 * every rank calls ocn_dist_create after ocn_init(local device). Halo transfers run on the library's communication stream
 * between two events: kernels launched afterwards overlap them; the host never synchronises (the reference calls sync_device!
 * before every MPI call, halo_communication.jl:181, distributed_transpose.jl:187). */
int ocn_dist_unique_id(void *id128);
int ocn_dist_create(ocn_dist_t *dist, const void *id128, int world, int rank);
/* The same architecture object over collectives the CALLER supplies -- the seam for MPI.jl in a Julia binder (the reference's
 * own transport), and what the tests use to run R ranks on one card. All buffers are device pointers; `stream` is the library's
 * compute stream (a hipStream_t): the callbacks must order their transfers behind the work already on it and make later work on
 * it wait for them (or synchronise). exchange_start/_wait: `count` doubles per side, what leaves through the west side arrives in the
 * west neighbour's east halo (MPI.Isend/Irecv! + Waitall, halo_communication.jl:300,326,164); all_to_all: piece r of `send`
 * (`count_per_rank` doubles) goes to rank r (MPI.Alltoallv!, distributed_transpose.jl:185-191); all_gather: rank r's `count` doubles
 * land at recv[r * count]; allreduce_max: host scalar in / out. Return 0 on success. */
typedef struct {
    int (*exchange_start)(void *user, const double *west_send, const double *east_send, double *west_recv, double *east_recv,
                          size_t count, void *stream);
    int (*exchange_wait)(void *user, void *stream);
    int (*all_to_all)(void *user, const double *send, double *recv, size_t count_per_rank, void *stream);
    int (*all_gather)(void *user, const double *send, double *recv, size_t count, void *stream);
    int (*allreduce_max)(void *user, double *value);
    void *user;
    /* pencil partitions only (may be NULL otherwise): one exchange with an explicit pair of peer ranks, complete in stream order when it
     * returns -- `lo_send` goes to peer_lo and lands in ITS hi_recv, `hi_send` goes to peer_hi and lands in its lo_recv */
    int (*exchange_peers)(void *user, int peer_lo, int peer_hi, const double *lo_send, const double *hi_send, double *lo_recv,
                          double *hi_recv, size_t count, void *stream);
    /* pencil transposes only (may be NULL otherwise): MPI.Alltoallv! with equal counts on a sub-communicator given as the list of its
     * members' world ranks (this rank included, in group order): chunk q of `send` (count doubles) goes to peers[q], chunk q of `recv`
     * comes from peers[q]; complete in stream order when it returns */
    int (*all_to_all_group)(void *user, const int *peers, int npeers, const double *send, double *recv, size_t count, void *stream);
} ocn_transport_t;
int ocn_dist_create_transport(ocn_dist_t *dist, const ocn_transport_t *transport, int world, int rank);
int ocn_dist_destroy(ocn_dist_t dist);
int ocn_dist_info(ocn_dist_t dist, int *world, int *rank, int *west, int *east);
/* what the TRANSPORT reports: kind 0 = the library's RCCL communicator (comm_ranks / comm_rank / device from ncclCommCount /
 * ncclCommUserRank / ncclCommCuDevice -- the ranks RCCL really connected, MPI.Comm_size / Comm_rank of the reference's communicator,
 * distributed_architectures.jl:262-263), 1 = caller-supplied collectives (the numbers given at creation) */
int ocn_dist_comm_info(ocn_dist_t dist, int *kind, int *comm_ranks, int *comm_rank, int *device);
/* MEASUREMENT / TEST ONLY: a communicator of ONE rank treats x as partitioned with itself as both neighbours, so the complete
 * N > 1 code path runs (and can be timed) on a one-GPU box; results equal the one-rank Periodic run. Set before model creation. */
int ocn_dist_set_self_loop(ocn_dist_t dist, int enabled);
/* the collectives themselves, for callers that orchestrate the stages on their own (send / recv of halo buffers,
 * halo_communication.jl:170-187; transposes, distributed_transpose.jl:185-191) */
int ocn_dist_exchange_start(ocn_dist_t dist, const double *west_send, const double *east_send, double *west_recv, double *east_recv,
                            size_t count);
int ocn_dist_exchange_wait(ocn_dist_t dist);
int ocn_dist_all_to_all(ocn_dist_t dist, const double *send, double *recv, size_t count_per_rank);
int ocn_dist_all_gather(ocn_dist_t dist, const double *send, double *recv, size_t count);
int ocn_dist_allreduce_max(ocn_dist_t dist, double *value);      /* synchronous */
int ocn_dist_barrier(ocn_dist_t dist);                           /* synchronous */
/* NonhydrostaticModel on a Distributed architecture: `local_grid` is this rank's slab (x topology OCN_CONNECTED when x is
 * partitioned: distributed_grids.jl:339-346), `Lx_global` the extent of the global domain along x. The returned handle is an
 * ocn_model_t: every ocn_model_* call works on it (set_option also takes "async_halos" -1 automatic / 0 / 1, "thin_halos",
 * "early_exchange", "strip_width"); ocn_model_time_step runs the partitioned RK3 step -- fill_halo_regions! with the x exchange
 * (halo_communication.jl:87-110), the interior / buffer split (Models/interleave_communication_and_computation.jl:9-67) or the
 * exchange started from make_pressure_correction!, and solve! of the distributed solvers
 * (distributed_fft_based_poisson_solver.jl:141-178, distributed_fft_tridiagonal_solver.jl:153-257) -- entirely inside the library. */
int ocn_dist_model_create(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global);
/* the same for an irregular partition (local_size, distributed_grids.jl:44-58: N / R columns per rank, the remainder on the last one;
 * or any `Sizes`): local_sizes[r] = Nx of rank r. Equal sizes take the distributed solvers; otherwise solve! gathers the source term
 * on every rank and runs the single-GPU solver on the global grid -- every slab layout works, memory and traffic grow with the
 * GLOBAL grid (a fallback, not the scaling path). */
int ocn_dist_model_create_sizes(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global,
                                const int *local_sizes);
/* ... and for a Bounded partitioned direction (global_x_topology = OCN_BOUNDED; OCN_PERIODIC: as above): insert_connected_topology
 * (distributed_grids.jl:339-346) -- the first rank's local grid is OCN_RIGHT_CONNECTED (wall on its west side), the last one's
 * OCN_LEFT_CONNECTED, the others OCN_CONNECTED. Boundary conditions and the advection scheme's wall fallbacks
 * (topologically_conditional_interpolation.jl:54-70) act on the wall side only; local_sizes may be NULL (equal slabs). The pressure
 * solve takes the gathered form on the global Bounded grid (cosine transform along x). */
int ocn_dist_model_create_partition(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global,
                                    const int *local_sizes, int global_x_topology);
/* Partition(Rx, Ry) pencils (distributed_architectures.jl:354-434: rank = ix * Ry + iy, periodic wrap of the four neighbours):
 * ocn_dist_set_layout fixes the layout of a communicator (ocn_dist_model_create_pencil calls it); the local grid is connected in x when
 * Rx > 1 (codes as above) and in y when Ry > 1: OCN_CONNECTED where the global y direction is Periodic, OCN_RIGHT_CONNECTED (first row
 * of ranks) / OCN_CONNECTED / OCN_LEFT_CONNECTED (last row) where global_y_topology = OCN_BOUNDED. sizes_x[Rx] / sizes_y[Ry]: slab
 * widths (NULL: equal).
 * Every fill makes two hops -- x, then y over the whole x extent -- so corners arrive without corner messages
 * (fill_corners!, halo_communication.jl:137-162); solve! is the gathered solve on the global grid (the reference's pencil transposes,
 * distributed_transpose.jl:12-15, are not built). */
int ocn_dist_set_layout(ocn_dist_t dist, int Rx, int Ry);
int ocn_dist_model_create_pencil(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global,
                                 double Ly_global, int Rx, int Ry, const int *sizes_x, const int *sizes_y, int global_x_topology,
                                 int global_y_topology);
/* ---- TransposableField and its transposes (pencil partitions; row f.4 of SURVEY.md 8) ----------------------------------------------
 * TransposableField(field_in, ComplexF64) (transposable_field.jl:49-105) for a field of GLOBAL size (Nx, Ny, Nz) on the communicator's
 * Partition(Rx, Ry) (ocn_dist_set_layout): complex (re, im interleaved) fields without halos, x fastest --
 * zfield (Nx/Rx, Ny/Ry, Nz), yfield (Nx/Rx, Ny, Nz/Ry), xfield (Nx, Ny/Rx, Nz/Ry); yfield is zfield when Ry = 1, xfield is yfield when
 * Rx = 1. Equal chunks as in the reference: Rx | Nx, Ry | Ny, Ry | Nz, Rx | Ny (distributed_fft_based_poisson_solver.jl:213-226). */
int ocn_transposable_create(ocn_transposable_t *field, ocn_dist_t dist, int Nx, int Ny, int Nz);
int ocn_transposable_destroy(ocn_transposable_t field);
/* device pointers and sizes of the three configurations (any output may be NULL) */
int ocn_transposable_fields(ocn_transposable_t field, double **zfield, double **yfield, double **xfield, int zsize[3], int ysize[3], int xsize[3]);
/* transpose_z_to_y! / transpose_y_to_x! / transpose_x_to_y! / transpose_y_to_z! (distributed_transpose.jl:25-95,185-191): pack kernel,
 * all-to-all inside the group of ranks that share ix (z <-> y) or iy (y <-> x), unpack kernel; bit-exact copies; no-ops on slabs (:12-15) */
int ocn_transpose_z_to_y(ocn_transposable_t field);
int ocn_transpose_y_to_x(ocn_transposable_t field);
int ocn_transpose_x_to_y(ocn_transposable_t field);
int ocn_transpose_y_to_z(ocn_transposable_t field);
int ocn_dist_model_max_abs_divergence(ocn_model_t model, double *value);    /* global maximum; synchronous */

#ifdef __cplusplus
}
#endif
#endif
