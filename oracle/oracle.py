"""ctypes front-end of the CPU oracle (oracle/ocn_oracle.c) + a numpy/decimal restatement of grid generation.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package never imports this module.

Grid generation follows reference src/Grids/grid_generation.jl:34-135 (citations inline). Parity status: see
ocn_oracle.h ("parity unpinned" vs Julia; pinned by the reference's data-free tests).
"""
import ctypes as C
import os
import subprocess
from decimal import Decimal, getcontext

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PERIODIC, BOUNDED, FLAT = 0, 1, 3
CENTER, FACE = 0, 1
LOC = {"u": (FACE, CENTER, CENTER), "v": (CENTER, FACE, CENTER), "w": (CENTER, CENTER, FACE), "c": (CENTER,) * 3}


def build(force=False):
    so = os.path.join(_HERE, "libocn_oracle.so")
    src = os.path.join(_HERE, "ocn_oracle.c")
    stamp = os.path.join(_HERE, ".build_arch")
    built_with_fma = os.path.exists(stamp) and "-mfma" in open(stamp).read()
    try:
        host_has_fma = " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        host_has_fma = False
    stale = not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale or (built_with_fma and not host_has_fma):     # never run -mfma code on a CPU without FMA
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def available_cpus():
    """CPU share of this process: affinity mask, cgroup quota and a cap of 16 (the GPU box's per-GPU share; a fresh box
    reports 256 hardware threads in its affinity mask but oversubscribing them stalls every OpenMP region)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("OCN_ORACLE_THREADS", "16"))))


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.oro_set_num_threads.argtypes = [C.c_int]
        L.oro_set_num_threads(available_cpus())
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.oro_grid_create.restype = vp
        L.oro_grid_create.argtypes = [ip, ip, ip, dp] + [dp] * 6
        L.oro_grid_destroy.argtypes = [vp]
        L.oro_parent_size.argtypes = [vp, ip, ip]
        L.oro_fill_halo_regions.argtypes = [vp, dp, ip, C.c_int]
        L.oro_fill_halo_regions_bcs.argtypes = [vp, dp, ip, C.POINTER(BC), C.c_int]
        L.oro_compute_flux_bcs.argtypes = [vp, dp, ip, C.POINTER(BC)]
        L.oro_model_set_bc.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.c_double]
        L.oro_model_set_bc_array.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, vp]
        L.oro_model_set_closure.argtypes = [vp, C.c_double, dp]
        L.oro_model_set_coriolis.argtypes = [vp, C.c_int, C.c_double]
        L.oro_add_fplane_coriolis.argtypes = [vp, C.c_double, dp, dp, dp, dp]
        L.oro_model_set_buoyancy.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.oro_update_hydrostatic_pressure.argtypes = [vp, C.c_int, dp, dp, C.c_double, C.c_double, C.c_double, dp]
        L.oro_add_hydrostatic_pressure_gradient.argtypes = [vp, dp, dp, dp]
        L.oro_add_closure_tendency.argtypes = [vp, C.c_int, dp, dp, dp, dp, C.c_double, dp, ip]
        L.oro_add_closure_tendency_field.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, dp, ip]
        L.oro_compute_amd_diffusivities.argtypes = [vp, C.c_double, dp, dp, dp, dp, C.POINTER(dp), C.c_int, dp, C.POINTER(dp)]
        L.oro_model_set_amd.argtypes = [vp, C.c_double, dp]
        L.oro_model_set_linear_flux_bc.argtypes = [vp, C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
        for n in ("oro_compute_Gu", "oro_compute_Gv", "oro_compute_Gw"):
            getattr(L, n).argtypes = [vp, dp, dp, dp, dp, ip]
        L.oro_compute_Gc.argtypes = [vp, dp, dp, dp, dp, dp, ip]
        L.oro_weno5_biased.restype = C.c_double
        L.oro_weno5_biased.argtypes = [dp, C.c_int]
        L.oro_weno3_biased.restype = C.c_double
        L.oro_weno3_biased.argtypes = [dp, C.c_int]
        L.oro_newton_div_f32.restype = C.c_double
        L.oro_newton_div_f32.argtypes = [C.c_double, C.c_double]
        L.oro_rk3_substep_field.argtypes = [vp, dp, ip, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp]
        L.oro_cache_tendencies.argtypes = [vp, dp, dp, ip]
        L.oro_compute_source_term.argtypes = [vp, dp, dp, dp, vp, C.c_int]
        L.oro_make_pressure_correction.argtypes = [vp, dp, dp, dp, dp]
        L.oro_scale_parent.argtypes = [vp, dp, ip, C.c_double]
        L.oro_poisson_create.restype = vp
        L.oro_poisson_create.argtypes = [vp, C.c_int]
        L.oro_poisson_destroy.argtypes = [vp]
        L.oro_poisson_rhs.restype = vp
        L.oro_poisson_rhs.argtypes = [vp]
        L.oro_poisson_solve.argtypes = [vp, dp]
        L.oro_batched_tridiagonal_solve_z.argtypes = [C.c_int] * 3 + [dp, dp, dp, vp, dp, vp]
        L.oro_poisson_eigenvalues.argtypes = [C.c_int, C.c_double, C.c_int, dp]
        L.oro_fft_line.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.oro_model_create.restype = vp
        L.oro_model_create.argtypes = [vp, C.c_int]
        L.oro_model_destroy.argtypes = [vp]
        L.oro_model_field.restype = dp
        L.oro_model_field.argtypes = [vp, C.c_char_p]
        L.oro_model_field_loc.argtypes = [vp, C.c_char_p, ip]
        L.oro_model_update_state.argtypes = [vp, C.c_int]
        L.oro_model_set_finalize.argtypes = [vp, C.c_int]
        L.oro_model_time_step.argtypes = [vp, C.c_double]
        L.oro_model_cell_advection_timescale.restype = C.c_double
        L.oro_model_cell_advection_timescale.argtypes = [vp]
        L.oro_model_time_step_ab2.argtypes = [vp, C.c_double, C.c_double, C.c_int]
        L.oro_model_time.restype = C.c_double
        L.oro_model_time.argtypes = [vp]
        L.oro_model_iteration.argtypes = [vp]
        L.oro_model_max_abs_divergence.restype = C.c_double
        L.oro_model_max_abs_divergence.argtypes = [vp]
        L.oro_set_num_threads.argtypes = [C.c_int]
        L.oro_adapt_advection_order.argtypes = [C.c_int] * 4
        L.oro_permute_index.argtypes = [C.c_int, C.c_int]
        L.oro_unpermute_index.argtypes = [C.c_int, C.c_int]
        L.oro_dct_makhoul.argtypes = [dp, C.c_int, C.c_int]
        L.oro_dct_direct.argtypes = [dp, C.c_int, C.c_int]
    return _LIB


class BC(C.Structure):
    """oro_bc: constant-valued boundary condition on one side"""
    _fields_ = [("kind", C.c_int), ("value", C.c_double), ("array", C.c_void_p)]


BC_KINDS = {"default": 0, "flux": 1, "value": 2, "gradient": 3, "open": 4}
SIDES = {"west": 0, "east": 1, "south": 2, "north": 3, "bottom": 4, "top": 5}


def _bcs(bcs):
    """dict side -> (kind, value)  ->  oro_bc[6]"""
    arr = (BC * 6)()
    arr._keep = []                 # array-valued conditions: (N_a, N_b) Fortran-ordered float64, kept alive with the struct
    for side, (kind, value) in (bcs or {}).items():
        arr[SIDES[side]].kind = BC_KINDS[kind]
        if isinstance(value, np.ndarray):
            a = np.asfortranarray(value, dtype=np.float64)
            arr._keep.append(a)
            arr[SIDES[side]].array = a.ctypes.data
        else:
            arr[SIDES[side]].value = float(value)
    return arr


def _dp(a):
    assert a.dtype == np.float64 and a.flags["F_CONTIGUOUS"] or a.ndim == 1
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i3(t):
    return (C.c_int * 3)(*t)


# --------------------------------------------------------------------------------------------------------------------
# grid generation (oracle restatement; decimal arithmetic with 80 digits stands in for Julia's 256-bit BigFloat)
# --------------------------------------------------------------------------------------------------------------------
def regular_spacing(interval, N):
    """grid_generation.jl:98-135: Δ = FT(BigFloat(L) / N), L = FT(c₂ - c₁)."""
    getcontext().prec = 80
    c1, c2 = Decimal(float(interval[0])), Decimal(float(interval[1]))
    assert c1 < c2, "must be an increasing interval!"
    L = c2 - c1
    return float(L / Decimal(N)), float(L)


def stretched_spacings(faces, N, H, bounded):
    """grid_generation.jl:34-95 for an explicit face vector of length N+1. Returns (L, Δᶜ, Δᶠ) with Δᶜ[idx], Δᶠ[idx]
    stored for idx = 1-H .. N+H+1 (array position idx-1+H); entries the reference does not define are padded by
    repeating the nearest defined value."""
    F_int = np.asarray(faces, dtype=np.float64)
    assert F_int.shape == (N + 1,) and np.all(np.diff(F_int) > 0)
    L = F_int[N] - F_int[0]
    if bounded:   # lower/upper_exterior_Δcoordᶠ(::BoundedTopology) :14,17
        dm = [F_int[1] - F_int[0]] * H
        dp = [F_int[-1] - F_int[-2]] * H
    else:         # :13,16
        dm = [F_int[N - H + i] - F_int[N - H + i - 1] for i in range(1, H + 1)]     # Fi[end-H+i] - Fi[end-H+i-1]
        dp = [F_int[i] - F_int[i - 1] for i in range(1, H + 1)]                     # Fi[i+1] - Fi[i]
    dp = dp[::-1]
    # F₋ = [c¹ - sum(Δᶠ₋[i:H])], F₊ = reverse([cᴺ⁺¹ + sum(Δᶠ₊[i:H])]) with Julia's left-to-right sum
    def jsum(v):
        s = v[0]
        for x in v[1:]:
            s = s + x
        return s
    Fm = [F_int[0] - jsum(dm[i:H]) for i in range(H)]
    Fp = [F_int[N] + jsum(dp[i:H]) for i in range(H)][::-1]
    F = np.concatenate([Fm, F_int, Fp])                       # N + 1 + 2H faces
    TC = N + 2 * H
    Cc = np.array([(F[i + 1] + F[i]) / 2 for i in range(TC)])
    dF = [Cc[i] - Cc[i - 1] for i in range(1, TC)]
    TF = N + 2 * H + (1 if bounded else 0)
    F = F[:TF]
    dC = [F[i + 1] - F[i] for i in range(TF - 1)]
    dF = [dF[0]] + dF + [dF[-1]]
    for i in range(len(dF) - 1, 0, -1):
        dF[i] = dF[i - 1]
    # reference indices: Δᶜ idx = 1-H .. (TF-1)-H ; Δᶠ idx = -H .. N+H
    n = N + 2 * H + 1
    dc = np.empty(n)
    df = np.empty(n)
    for pos in range(n):
        idx = pos + 1 - H
        pc = idx - (1 - H)
        dc[pos] = dC[min(pc, len(dC) - 1)]
        pf = idx - (-H)
        df[pos] = dF[min(pf, len(dF) - 1)]
    return float(L), dc, df


class Grid:
    """RectilinearGrid (x, y regular; z regular or stretched-Bounded). reference: Grids/rectilinear_grid.jl:264-291"""

    def __init__(self, size, halo=None, topology=(PERIODIC, PERIODIC, PERIODIC),
                 x=(0.0, 1.0), y=(0.0, 1.0), z=(0.0, 1.0)):
        self.N = tuple(int(n) for n in size)
        self.topo = tuple(int(t) for t in topology)
        if halo is None:      # validate_halo(..., ::Nothing) (Grids/input_validation.jl:71-77): min(3, size)
            halo = tuple(min(3, n) for n in self.N)
        # buffers of the per-direction schemes NonhydrostaticModel(advection = WENO()) ends up with
        self.B = tuple(lib().oro_adapt_advection_order(2, 3, self.N[d], self.topo[d]) for d in range(3))
        # Flat directions: one cell, no halo, unit spacing and extent (Grids/grid_utils.jl, spacings_and_areas_and_volumes.jl:123)
        assert all(self.N[d] == 1 for d in range(3) if self.topo[d] == FLAT), "a Flat direction has size 1"
        self.H = tuple(0 if self.topo[d] == FLAT else int(h) for d, h in enumerate(halo))
        self.L = [0.0] * 3
        self.dc, self.df = [None] * 3, [None] * 3
        for d, coord in enumerate((x, y, z)):
            n = self.N[d] + 2 * self.H[d] + 1
            if self.topo[d] == FLAT:
                self.L[d], self.dc[d], self.df[d] = 1.0, np.ones(n), np.ones(n)
            elif isinstance(coord, tuple) and len(coord) == 2 and np.isscalar(coord[0]):
                delta, L = regular_spacing(coord, self.N[d])
                self.L[d] = L
                self.dc[d] = np.full(n, delta)
                self.df[d] = np.full(n, delta)
            else:
                self.L[d], self.dc[d], self.df[d] = stretched_spacings(coord, self.N[d], self.H[d],
                                                                       self.topo[d] == BOUNDED)
        self.handle = lib().oro_grid_create(_i3(self.N), _i3(self.H), _i3(self.topo), (C.c_double * 3)(*self.L),
                                            _dp(self.dc[0]), _dp(self.df[0]), _dp(self.dc[1]), _dp(self.df[1]),
                                            _dp(self.dc[2]), _dp(self.df[2]))

    def parent_size(self, loc):
        return tuple(self.N[d] + 2 * self.H[d] + (1 if (loc[d] == FACE and self.topo[d] == BOUNDED) else 0)
                     for d in range(3))

    def zeros(self, loc):
        return np.zeros(self.parent_size(loc), dtype=np.float64, order="F")

    def interior(self, a, loc):
        sl = []
        for d in range(3):
            n = self.N[d] + (1 if (loc[d] == FACE and self.topo[d] == BOUNDED) else 0)
            sl.append(slice(self.H[d], self.H[d] + n))
        return a[tuple(sl)]

    def interior_cells(self, a):
        return a[tuple(slice(self.H[d], self.H[d] + self.N[d]) for d in range(3))]

    # ---- kernels ----
    def fill_halo_regions(self, a, loc, fill_open_bcs=True, bcs=None):
        if bcs is None:
            lib().oro_fill_halo_regions(self.handle, _dp(a), _i3(loc), int(fill_open_bcs))
        else:
            lib().oro_fill_halo_regions_bcs(self.handle, _dp(a), _i3(loc), _bcs(bcs), int(fill_open_bcs))

    def compute_flux_bcs(self, G, loc, bcs):
        lib().oro_compute_flux_bcs(self.handle, _dp(G), _i3(loc), _bcs(bcs))

    def compute_G(self, which, u, v, w, G, c=None, rng=None):
        r = (C.c_int * 6)(*rng) if rng is not None else None
        if which == "c":
            lib().oro_compute_Gc(self.handle, _dp(u), _dp(v), _dp(w), _dp(c), _dp(G), r)
        else:
            getattr(lib(), "oro_compute_G" + which)(self.handle, _dp(u), _dp(v), _dp(w), _dp(G), r)

    def rk3_substep(self, U, loc, dt, gamma, zeta, Gn, Gm):
        lib().oro_rk3_substep_field(self.handle, _dp(U), _i3(loc), dt, gamma, 0.0 if zeta is None else zeta,
                                    0 if zeta is None else 1, _dp(Gn), _dp(Gm))

    def source_term(self, u, v, w, weight_by_dz=False):
        rhs = np.zeros(self.N, dtype=np.complex128, order="F")
        lib().oro_compute_source_term(self.handle, _dp(u), _dp(v), _dp(w), rhs.ctypes.data, int(weight_by_dz))
        return rhs

    def pressure_correct(self, u, v, w, p):
        lib().oro_make_pressure_correction(self.handle, _dp(u), _dp(v), _dp(w), _dp(p))


class PoissonSolver:
    def __init__(self, grid, kind):
        self.grid, self.kind = grid, kind
        self.handle = lib().oro_poisson_create(grid.handle, kind)
        n = grid.N[0] * grid.N[1] * grid.N[2]
        buf = (C.c_double * (2 * n)).from_address(lib().oro_poisson_rhs(self.handle))
        self.rhs = np.frombuffer(buf, dtype=np.complex128).reshape(grid.N, order="F")

    def solve(self, phi):
        lib().oro_poisson_solve(self.handle, _dp(phi))

    def __del__(self):
        try:
            lib().oro_poisson_destroy(self.handle)
        except Exception:
            pass


class Model:
    """NonhydrostaticModel(grid; advection=WENO(), tracers, timestepper=:RungeKutta3) on the CPU oracle."""

    def __init__(self, grid, ntracers=2):
        self.grid, self.ntracers = grid, ntracers
        self.handle = lib().oro_model_create(grid.handle, ntracers)

    def names(self):
        return ["u", "v", "w"] + ["c%d" % t for t in range(self.ntracers)]

    def field(self, name):
        """numpy view (parent array with halos, Fortran order) of a model field."""
        loc = (C.c_int * 3)()
        lib().oro_model_field_loc(self.handle, name.encode(), loc)
        shape = self.grid.parent_size(tuple(loc))
        ptr = lib().oro_model_field(self.handle, name.encode())
        n = int(np.prod(shape))
        buf = (C.c_double * n).from_address(C.addressof(ptr.contents))
        return np.frombuffer(buf, dtype=np.float64).reshape(shape, order="F")

    def loc(self, name):
        loc = (C.c_int * 3)()
        lib().oro_model_field_loc(self.handle, name.encode(), loc)
        return tuple(loc)

    def set_buoyancy_tracer(self, b_index=0):
        """buoyancy = BuoyancyTracer(): tracer c<b_index> is the buoyancy"""
        assert lib().oro_model_set_buoyancy(self.handle, 1, b_index, 0, 0.0, 0.0, 0.0) == 0

    def set_seawater_buoyancy(self, T_index=0, S_index=1, g=9.80665, alpha=1.67e-4, beta=7.80e-4):
        """buoyancy = SeawaterBuoyancy(equation_of_state = LinearEquationOfState(α, β)) with the reference's default constants"""
        assert lib().oro_model_set_buoyancy(self.handle, 2, T_index, S_index, g, alpha, beta) == 0

    def set_coriolis(self, f):
        """coriolis = FPlane(f = f)"""
        lib().oro_model_set_coriolis(self.handle, 1, float(f))

    def set_closure(self, nu=0.0, kappa=0.0):
        """closure = ScalarDiffusivity(ν = nu, κ = kappa) -- kappa a number or one value per tracer"""
        k = np.atleast_1d(np.asarray(kappa, dtype=np.float64))
        if k.size == 1:
            k = np.full(max(self.ntracers, 1), float(k[0]))
        k = np.ascontiguousarray(k)
        lib().oro_model_set_closure(self.handle, float(nu), k.ctypes.data_as(C.POINTER(C.c_double)))

    def set_amd(self, C=1 / 3, Cnu=None, Ckappa=None):
        """closure = AnisotropicMinimumDissipation(C = C, Cν = Cnu, Cκ = Ckappa): Ckappa a number or one value per tracer.
        The eddy coefficients are the fields "nu_e", "kappa_e0", ..."""
        Cnu = C if Cnu is None else Cnu
        k = np.atleast_1d(np.asarray(C if Ckappa is None else Ckappa, dtype=np.float64))
        if k.size == 1:
            k = np.full(max(self.ntracers, 1), float(k[0]))
        k = np.ascontiguousarray(k)
        import ctypes                      # the keyword `C` (the reference's name) shadows the module alias here
        if lib().oro_model_set_amd(self.handle, float(Cnu), k.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) != 0:
            raise ValueError("AnisotropicMinimumDissipation needs a grid without Flat directions")

    def set_linear_flux_bc(self, name, side, a, b, dep):
        """name.side = FluxBoundaryCondition((x, y, t, φ, p) -> a + b φ, field_dependencies = dep)"""
        if lib().oro_model_set_linear_flux_bc(self.handle, name.encode(), SIDES[side], float(a), float(b), dep.encode()) != 0:
            raise ValueError(f"invalid field-dependent flux condition on the {side} side of {name} (dependency {dep})")

    def set_bc(self, name, side, kind, value=0.0):
        """field boundary condition with a constant value or a 2-D array of per-point values: kind in flux | value | gradient | open |
        default"""
        if isinstance(value, np.ndarray):
            a = np.asfortranarray(value, dtype=np.float64)
            self._bc_arrays = getattr(self, "_bc_arrays", []) + [a]           # borrowed by the C model
            rc = lib().oro_model_set_bc_array(self.handle, name.encode(), SIDES[side], BC_KINDS[kind], a.ctypes.data)
        else:
            rc = lib().oro_model_set_bc(self.handle, name.encode(), SIDES[side], BC_KINDS[kind], float(value))
        if rc != 0:
            raise ValueError(f"invalid boundary condition {kind} on the {side} side of {name}")

    def set(self, enforce_incompressibility=True, **fields):
        for name, val in fields.items():
            a = self.field(name)
            self.grid.interior(a, self.loc(name))[...] = val
        lib().oro_model_set_finalize(self.handle, int(enforce_incompressibility))

    def update_state(self, compute_tendencies=True):
        lib().oro_model_update_state(self.handle, int(compute_tendencies))

    def time_step(self, dt):
        lib().oro_model_time_step(self.handle, float(dt))

    def cell_advection_timescale(self):
        return lib().oro_model_cell_advection_timescale(self.handle)

    def time_step_ab2(self, dt, chi=0.1, euler=False):
        """time_step!(model::AbstractModel{<:QuasiAdamsBashforth2TimeStepper}, Δt; euler)"""
        lib().oro_model_time_step_ab2(self.handle, float(dt), float(chi), int(euler))

    @property
    def time(self):
        return lib().oro_model_time(self.handle)

    @property
    def iteration(self):
        return lib().oro_model_iteration(self.handle)

    def max_abs_divergence(self):
        return lib().oro_model_max_abs_divergence(self.handle)

    def __del__(self):
        try:
            lib().oro_model_destroy(self.handle)
        except Exception:
            pass


def compute_amd_diffusivities(grid, Cnu, Ckappa, u, v, w, tracers):
    """νₑ and κₑ[t] (parent arrays with halos, interiors filled) from parent arrays u, v, w, tracers with filled halos"""
    dp = C.POINTER(C.c_double)
    arr = lambda a: np.asfortranarray(a, dtype=np.float64)          # noqa: E731
    u, v, w = arr(u), arr(v), arr(w)
    tracers = [arr(t) for t in tracers]
    shape = grid.parent_size((CENTER, CENTER, CENTER))
    nu = np.zeros(shape, order="F")
    kap = [np.zeros(shape, order="F") for _ in tracers]
    ck = np.ascontiguousarray(np.asarray(Ckappa, dtype=np.float64).reshape(-1)) if tracers else np.zeros(1)
    tp = (dp * max(len(tracers), 1))(*[t.ctypes.data_as(dp) for t in tracers])
    kp = (dp * max(len(tracers), 1))(*[k.ctypes.data_as(dp) for k in kap])
    lib().oro_compute_amd_diffusivities(grid.handle, float(Cnu), ck.ctypes.data_as(dp), u.ctypes.data_as(dp), v.ctypes.data_as(dp),
                                        w.ctypes.data_as(dp), tp, len(tracers), nu.ctypes.data_as(dp), kp)
    return nu, kap
