"""CPU tests that PIN THE ORACLE with the reference's own data-free tests and docstring KATs -- the oracle is
"parity unpinned" against Julia output (no Julia here, regression data are remote DataDeps), so these properties are what
anchors it. Each test names the reference test it restates."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import tanh_faces

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, B = 0, 1


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ---------------------------------------------------------------------------------------------------------------------
# coefficients (Advection/reconstruction_coefficients.jl docstrings)
# ---------------------------------------------------------------------------------------------------------------------
def test_coefficient_kats_and_header_is_current():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_coefficients.py")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert res.stdout == open(os.path.join(ROOT, "include", "ocn_weno_coeffs.h")).read()


def test_rk3_stage_fractions_sum_to_one():
    """runge_kutta_3.jl:69-78,176-177"""
    g1, g2, g3, z2, z3 = 8 / 15, 5 / 12, 3 / 4, -17 / 60, -5 / 12
    assert abs(g1 + (g2 + z2) + (g3 + z3) - 1) < 1e-15


# ---------------------------------------------------------------------------------------------------------------------
# WENO reconstruction (Advection/weno_interpolants.jl)
# ---------------------------------------------------------------------------------------------------------------------
def test_newton_div_is_a_double_precision_quotient(oracle):
    rng = np.random.default_rng(0)
    a, b = rng.random(1000) * 10, rng.random(1000) * 10 + 1e-8
    got = np.array([oracle.lib().oro_newton_div_f32(x, y) for x, y in zip(a, b)])
    assert np.max(np.abs(got - a / b) / (a / b)) < 1e-13        # Float32 reciprocal + one Newton step (newton_div.jl:8-20)


def test_weno5_exact_for_quadratics_and_mirror_symmetric(oracle):
    L = oracle.lib()
    x = np.arange(-3, 3) + 0.5
    S = (1 + 2 * x + 3 * (x * x + 1 / 12)).astype(np.float64)      # cell averages of 1 + 2x + 3x^2, face at x = 0
    for left in (1, 0):
        assert abs(L.oro_weno5_biased(dptr(S), left) - 1.0) < 1e-14
    rng = np.random.default_rng(1)
    for _ in range(100):
        S = rng.standard_normal(6)
        R = np.ascontiguousarray(S[::-1])
        # right-biased reconstruction == left-biased reconstruction of the mirrored stencil (S₀₃..S₂₃, :435-437)
        assert L.oro_weno5_biased(dptr(S), 0) == L.oro_weno5_biased(dptr(R), 1)
    S4 = rng.standard_normal(4)
    assert L.oro_weno3_biased(dptr(S4), 0) == L.oro_weno3_biased(dptr(np.ascontiguousarray(S4[::-1])), 1)


def test_weno5_fifth_order_on_smooth_data(oracle):
    L = oracle.lib()
    errs = []
    for n in (16, 32, 64, 128):
        h = 1.0 / n
        xs = (np.arange(-3, 3) + 0.5) * h + 0.3
        S = (np.cos(2 * np.pi * (xs - h / 2)) - np.cos(2 * np.pi * (xs + h / 2))) / (2 * np.pi * h)   # cell averages of sin
        errs.append(abs(L.oro_weno5_biased(dptr(S), 1) - np.sin(2 * np.pi * 0.3)))
    rates = np.log2(np.array(errs[:-1]) / np.array(errs[1:]))
    assert rates[-1] > 4.7, rates          # expected 2K-1 = 5 (validation/convergence_tests/one_dimensional_advection_schemes.jl:56-58)


# ---------------------------------------------------------------------------------------------------------------------
# halo regions (test/test_halo_regions.jl:22-41)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [(8, 8, 8), (3, 8, 8), (8, 3, 8), (8, 8, 3)])
def test_periodic_and_no_flux_halos_are_exact(oracle, N):
    rng = np.random.default_rng(2)
    for topo in ((P, P, P), (P, P, B), (B, B, B)):
        g = oracle.Grid(N, topology=topo)
        a = g.zeros(oracle.LOC["c"])
        g.interior_cells(a)[...] = rng.standard_normal(N)
        g.fill_halo_regions(a, oracle.LOC["c"])
        H = 3
        for d in range(3):
            sl = lambda s: tuple(s if q == d else slice(H, -H) for q in range(3))   # noqa: E731
            if topo[d] == P:
                assert np.array_equal(a[sl(slice(0, H))], a[sl(slice(N[d], N[d] + H))])
                assert np.array_equal(a[sl(slice(N[d] + H, N[d] + 2 * H))], a[sl(slice(H, 2 * H))])
            else:       # no-flux: c[0] == c[1], c[N+1] == c[N]
                assert np.array_equal(a[sl(slice(H - 1, H))], a[sl(slice(H, H + 1))])
                assert np.array_equal(a[sl(slice(N[d] + H, N[d] + H + 1))], a[sl(slice(N[d] + H - 1, N[d] + H))])


def test_impenetrable_walls_and_fill_open_bcs_flag(oracle):
    g = oracle.Grid((6, 5, 4), topology=(P, P, B))
    w = g.zeros(oracle.LOC["w"])
    assert w.shape == (12, 11, 11)                 # Face field on a Bounded dim has N+1 interior points (grid_utils.jl:66-72)
    w[...] = 1.0
    g.fill_halo_regions(w, oracle.LOC["w"], fill_open_bcs=False)
    assert np.all(w[3:-3, 3:-3, 3] == 1.0)
    g.fill_halo_regions(w, oracle.LOC["w"], fill_open_bcs=True)
    assert np.all(w[:, :, 3] == 0.0) and np.all(w[:, :, 3 + 4] == 0.0)      # wall faces k = 1 and k = Nz+1, x/y halos included


def test_value_gradient_and_open_halo_fills(oracle):
    """fill_halo_regions_value_gradient.jl:7-119: ONE halo cell by linear extrapolation through the boundary face, so that
    the face average equals a Value condition and the face difference a Gradient condition; fill_halo_regions_open.jl:2-7
    puts an Open condition on the wall-normal component (test/test_boundary_conditions_integration.jl exercises the same
    kernels through models)"""
    N = (6, 5, 4)
    zf = tanh_faces(N[2])
    g = oracle.Grid(N, topology=(B, B, B), z=zf)
    rng = np.random.default_rng(3)
    loc = oracle.LOC["c"]
    c = np.asfortranarray(rng.standard_normal(g.parent_size(loc)))
    before = c.copy()
    bcs = {"west": ("value", 1.5), "east": ("gradient", -0.7), "south": ("gradient", 0.3), "north": ("value", -2.0),
           "bottom": ("value", 0.25), "top": ("gradient", 4.0)}
    g.fill_halo_regions(c, loc, True, bcs=bcs)
    I = (slice(3, 3 + N[0]), slice(3, 3 + N[1]), slice(3, 3 + N[2]))
    dx, dy = 1.0 / N[0], 1.0 / N[1]
    dzf_bot, dzf_top = g.df[2][3], g.df[2][3 + N[2]]            # Δzᶠ at k = 1 and k = Nz + 1
    assert np.allclose((c[2, I[1], I[2]] + c[3, I[1], I[2]]) / 2, 1.5, rtol=0, atol=1e-14)
    assert np.allclose((c[3 + N[0], I[1], I[2]] - c[2 + N[0], I[1], I[2]]) / dx, -0.7, rtol=0, atol=1e-13)
    assert np.allclose((c[I[0], 3, I[2]] - c[I[0], 2, I[2]]) / dy, 0.3, rtol=0, atol=1e-13)
    assert np.allclose((c[I[0], 3 + N[1], I[2]] + c[I[0], 2 + N[1], I[2]]) / 2, -2.0, rtol=0, atol=1e-14)
    assert np.allclose((c[I[0], I[1], 2] + c[I[0], I[1], 3]) / 2, 0.25, rtol=0, atol=1e-14)
    assert np.allclose((c[I[0], I[1], 3 + N[2]] - c[I[0], I[1], 2 + N[2]]) / dzf_top, 4.0, rtol=0, atol=1e-12)
    assert dzf_bot > 0
    # only ONE halo cell per side is written, and only over the interior extent of the other two dimensions
    assert np.array_equal(c[:2], before[:2]) and np.array_equal(c[I], before[I]) and np.array_equal(c[2, :3], before[2, :3])
    # Open condition on w: wall-normal component takes the prescribed value when fill_open_bcs, untouched otherwise
    w = g.zeros(oracle.LOC["w"])
    w[...] = 1.0
    g.fill_halo_regions(w, oracle.LOC["w"], False, bcs={"top": ("open", 0.125)})
    assert np.all(w[3:-3, 3:-3, 3 + N[2]] == 1.0)
    g.fill_halo_regions(w, oracle.LOC["w"], True, bcs={"top": ("open", 0.125)})
    assert np.all(w[3:-3, 3:-3, 3 + N[2]] == 0.125) and np.all(w[3:-3, 3:-3, 3] == 0.0)


@pytest.mark.parametrize("topo,names,sides", [
    ((P, B, B), ("u", "c0"), ("north", "south", "top", "bottom")),
    ((B, P, B), ("v", "c0"), ("east", "west", "top", "bottom")),
    ((B, B, P), ("w", "c0"), ("east", "west", "north", "south")),
])
def test_flux_boundary_condition_budget(oracle, topo, names, sides):
    """test_nonhydrostatic_flux_budget (test/test_boundary_conditions_integration.jl:28-52,309-358): a Flux condition of
    ±π on one side of a field that starts at 0 gives mean(ϕ) = flux * t / L after one RK3 step with Δt = 1"""
    Lx, Ly, Lz = 0.3, 0.4, 0.5
    L = {"east": Lx, "west": Lx, "north": Ly, "south": Ly, "top": Lz, "bottom": Lz}
    for name in names:
        for side in sides:
            g = oracle.Grid((4, 4, 4), topology=topo, x=(0.0, Lx), y=(0.0, Ly), z=(0.0, Lz))
            m = oracle.Model(g, 1)
            direction = 1 if side in ("west", "south", "bottom") else -1
            m.set_bc(name, side, "flux", np.pi * direction)
            m.set(**{n: 0.0 for n in m.names()})
            m.time_step(1.0)
            mean = g.interior(m.field(name), m.loc(name)).mean()
            assert abs(mean - np.pi * m.time / L[side]) < 1e-12 * np.pi / L[side], (name, side, mean)


def test_boundary_condition_validation(oracle):
    g = oracle.Grid((4, 4, 4), topology=(P, P, B))
    m = oracle.Model(g, 1)
    with pytest.raises(ValueError):
        m.set_bc("c0", "west", "flux", 1.0)        # Periodic side
    with pytest.raises(ValueError):
        m.set_bc("w", "top", "value", 1.0)         # wall-normal component takes Open conditions only
    with pytest.raises(ValueError):
        m.set_bc("u", "top", "open", 1.0)
    m.set_bc("u", "top", "flux", 1.0)


# ---------------------------------------------------------------------------------------------------------------------
# Poisson solvers (test/test_poisson_solvers.jl:58-108, dependencies_for_poisson_solvers.jl:111-173,195-220)
# ---------------------------------------------------------------------------------------------------------------------
def _laplacian(g, p):
    """∇²p on the interior with the grid's own spacings: (δ(δp/Δᶠ))/Δᶜ per direction"""
    H = 3
    out = 0.0
    for ax in range(3):
        n = g.N[ax]
        other = tuple(slice(H, -H) if q != ax else slice(None) for q in range(3))
        a = p[other]
        lo, mid, hi = (np.take(a, range(H - 1 + s, H - 1 + s + n), axis=ax) for s in (0, 1, 2))
        shp = [1, 1, 1]
        shp[ax] = n
        dc = np.asarray(g.dc[ax][H:H + n]).reshape(shp)          # Δᶜ at cells 1..N   (array position idx-1+H)
        dfl = np.asarray(g.df[ax][H:H + n]).reshape(shp)         # Δᶠ at faces 1..N
        dfh = np.asarray(g.df[ax][H + 1:H + 1 + n]).reshape(shp)  # Δᶠ at faces 2..N+1
        out = out + ((hi - mid) / dfh - (mid - lo) / dfl) / dc
    return out


def _random_divergence_free_test(oracle, g, solver_kind):
    """∇²ϕ ≈ ∇·U for random U with impenetrable walls"""
    rng = np.random.default_rng(3)
    u, v, w = (g.zeros(oracle.LOC[k]) for k in "uvw")
    for a, k in ((u, "u"), (v, "v"), (w, "w")):
        view = g.interior(a, oracle.LOC[k])
        view[...] = rng.standard_normal(view.shape)
        g.fill_halo_regions(a, oracle.LOC[k])
    R = g.source_term(u, v, w, weight_by_dz=False).real.copy()
    s = oracle.PoissonSolver(g, solver_kind)
    s.rhs[...] = g.source_term(u, v, w, weight_by_dz=(solver_kind == 1))
    p = g.zeros(oracle.LOC["c"])
    s.solve(p)
    g.fill_halo_regions(p, oracle.LOC["c"])
    lap = _laplacian(g, p)
    assert np.allclose(lap, R, atol=1e-10 * max(1.0, np.abs(R).max())), np.abs(lap - R).max()


@pytest.mark.parametrize("topo", [(P, P, P), (P, P, B), (P, B, B), (B, B, B), (B, P, P), (P, B, P)])
@pytest.mark.parametrize("N", [(7, 7, 7), (16, 16, 16), (11, 16, 7)])
def test_fft_poisson_solver_divergence_free(oracle, topo, N):
    _random_divergence_free_test(oracle, oracle.Grid(N, topology=topo), 0)


@pytest.mark.parametrize("N", [(8, 8, 16), (7, 11, 9)])
def test_fourier_tridiagonal_solver_on_stretched_grid(oracle, N):
    g = oracle.Grid(N, topology=(P, P, B), z=tanh_faces(N[2]))
    _random_divergence_free_test(oracle, g, 1)


def test_fourier_tridiagonal_equals_fft_solver_on_regular_bounded_z(oracle):
    """the product routes z-Bounded grids through the tridiagonal solver: same discrete operator"""
    rng = np.random.default_rng(4)
    g = oracle.Grid((8, 8, 8), topology=(P, P, B), z=(-1.0, 0.0))
    R = rng.standard_normal(g.N)
    R -= R.mean()
    s0, s1 = oracle.PoissonSolver(g, 0), oracle.PoissonSolver(g, 1)
    s0.rhs[...] = R
    s1.rhs[...] = R * g.dc[2][0]
    p0, p1 = g.zeros(oracle.LOC["c"]), g.zeros(oracle.LOC["c"])
    s0.solve(p0)
    s1.solve(p1)
    assert np.allclose(p0, p1, atol=1e-13)


def test_poisson_second_order_convergence(oracle):
    """analytic cosines, rate ≈ 2 (dependencies_for_poisson_solvers.jl:135-173)"""
    errs = []
    for n in (16, 32, 64):
        g = oracle.Grid((n, n, n), topology=(P, P, B), x=(0.0, 2 * np.pi), y=(0.0, 2 * np.pi), z=(0.0, np.pi))
        h = [g.dc[d][0] for d in range(3)]
        x = (np.arange(n) + 0.5) * h[0]
        y = (np.arange(n) + 0.5) * h[1]
        z = (np.arange(n) + 0.5) * h[2]
        X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
        phi = np.cos(2 * X) * np.sin(3 * Y) * np.cos(2 * Z)
        s = oracle.PoissonSolver(g, 0)
        s.rhs[...] = -(4 + 9 + 4) * phi
        p = g.zeros(oracle.LOC["c"])
        s.solve(p)
        errs.append(np.abs(g.interior_cells(p) - phi).max())
    rates = np.log2(np.array(errs[:-1]) / np.array(errs[1:]))
    assert np.all(np.abs(rates - 2) < 0.1), rates


def test_batched_tridiagonal_solver_vs_dense(oracle):
    """test/test_batched_tridiagonal_solver.jl:7-93"""
    rng = np.random.default_rng(5)
    Nx, Ny, Nz = 3, 4, 12
    a, c = rng.random(Nz - 1), rng.random(Nz - 1)
    b = np.asfortranarray(3 + rng.random((Nx, Ny, Nz)))
    f = np.asfortranarray(rng.standard_normal((Nx, Ny, Nz)) + 1j * rng.standard_normal((Nx, Ny, Nz)))
    t = np.zeros((Nx, Ny, Nz), order="F")
    phi = np.zeros((Nx, Ny, Nz), dtype=np.complex128, order="F")
    oracle.lib().oro_batched_tridiagonal_solve_z(Nx, Ny, Nz, dptr(a), dptr(b), dptr(c), f.ctypes.data, dptr(t), phi.ctypes.data)
    for i in range(Nx):
        for j in range(Ny):
            M = np.diag(b[i, j]) + np.diag(a, -1) + np.diag(c, 1)
            assert np.allclose(phi[i, j], np.linalg.solve(M, f[i, j]), rtol=1e-12, atol=1e-13)


# ---------------------------------------------------------------------------------------------------------------------
# time stepping (test/test_time_stepping.jl:124-199,432-460; test/test_dynamics.jl:216-261)
# ---------------------------------------------------------------------------------------------------------------------
def _random_model(oracle, g, seed=6):
    rng = np.random.default_rng(seed)
    m = oracle.Model(g, 2)
    vals = {n: rng.standard_normal(g.interior(m.field(n), m.loc(n)).shape) for n in m.names()}
    m.set(**vals)
    return m


@pytest.mark.parametrize("kind", ["regular", "stretched", "bounded-regular"])
def test_incompressibility_after_1_and_10_rk3_steps(oracle, kind):
    N = (16, 16, 16)
    if kind == "regular":
        g = oracle.Grid(N)
    elif kind == "stretched":
        g = oracle.Grid(N, topology=(P, P, B), z=tanh_faces(16))
    else:
        g = oracle.Grid(N, topology=(P, P, B), z=(-1.0, 0.0))
    m = _random_model(oracle, g)
    assert m.max_abs_divergence() < 5e-8
    for nsteps in (1, 9):
        for _ in range(nsteps):
            m.time_step(1e-3)
        assert m.max_abs_divergence() < 5e-8
    assert m.iteration == 10 and abs(m.time - 1e-2) < 1e-15


def test_tracer_volume_integral_is_conserved_in_bounded_channel(oracle):
    """test/test_time_stepping.jl:165-199"""
    g = oracle.Grid((12, 10, 8), topology=(P, B, B), z=tanh_faces(8))
    m = _random_model(oracle, g, seed=7)
    vol = np.asarray(g.dc[2][3:3 + 8]).reshape(1, 1, 8)

    def total(name):
        return float((g.interior_cells(m.field(name)) * vol).sum())
    before = [total("c0"), total("c1")]
    for _ in range(10):
        m.time_step(5e-4)
    after = [total("c0"), total("c1")]
    assert np.allclose(before, after, rtol=0, atol=1e-11 * 12 * 10 * 8)


def test_taylor_green_vortex_is_steady(oracle):
    """2-D Taylor-Green flow is an exact steady Euler solution (closure = nothing); cf. test/test_dynamics.jl:216-261"""
    n = 32
    g = oracle.Grid((n, n, 4), x=(0.0, 2 * np.pi), y=(0.0, 2 * np.pi), z=(0.0, 1.0))
    h = g.dc[0][0]
    xf, xc = np.arange(n) * h, (np.arange(n) + 0.5) * h
    u0 = (np.cos(xf)[:, None, None] * np.sin(xc)[None, :, None]) * np.ones((1, 1, 4))
    v0 = (-np.sin(xc)[:, None, None] * np.cos(xf)[None, :, None]) * np.ones((1, 1, 4))
    m = oracle.Model(g, 1)
    m.set(u=u0, v=v0, w=0 * u0, c0=0 * u0)
    for _ in range(20):
        m.time_step(0.1 * h)
    u = g.interior_cells(m.field("u"))
    assert np.abs(u - u0).max() < 5e-4 and np.abs(g.interior_cells(m.field("w"))).max() < 1e-12


def test_advection_order_and_directional_symmetry(oracle):
    """validation/convergence_tests/one_dimensional_advection_schemes.jl:21-36,108-118: errors of x-, y-, z-oriented
    advection agree and converge at high order"""
    def run(n, axis):
        size = [4, 4, 4]
        size[axis] = n
        ext = [(0.0, 4.0 / n)] * 3
        ext[axis] = (0.0, 1.0)
        g = oracle.Grid(tuple(size), x=ext[0], y=ext[1], z=ext[2])
        h = 1.0 / n
        s = (np.arange(n) + 0.5) * h
        shp = [1, 1, 1]
        shp[axis] = n
        c0 = np.exp(-((s - 0.5) ** 2) / 0.01).reshape(shp) * np.ones(size)
        m = oracle.Model(g, 1)
        vel = {"u": 0.0, "v": 0.0, "w": 0.0}
        vel["uvw"[axis]] = 1.0
        m.set(u=vel["u"] + 0 * c0, v=vel["v"] + 0 * c0, w=vel["w"] + 0 * c0, c0=c0)
        dt, nsteps = 0.01 * h, 10
        for _ in range(nsteps):
            m.time_step(dt)
        exact = np.exp(-((s - 0.5 - dt * nsteps) ** 2) / 0.01).reshape(shp) * np.ones(size)
        return np.abs(g.interior_cells(m.field("c0")) - exact).max()
    e = {ax: [run(n, ax) for n in (32, 64)] for ax in range(3)}
    for ax in (1, 2):
        assert np.allclose(e[ax], e[0], rtol=1e-9), (e[0], e[ax])
    assert np.log2(e[0][0] / e[0][1]) > 3.5, e[0]


# ---------------------------------------------------------------------------------------------------------------------
# Flat directions (Grids/grid_utils.jl; Operators/difference_operators.jl:30-49; Advection/flat_advective_fluxes.jl:9-50)
# ---------------------------------------------------------------------------------------------------------------------
FLAT = 3


def test_flat_one_dimensional_advection_as_the_reference_runs_it(oracle):
    """validation/convergence_tests/src/OneDimensionalGaussianAdvectionDiffusion.jl:14-40 builds (Nx, 1, 1) grids with Flat y, z:
    same errors as the x-oriented run on a 3-D periodic grid (the Flat shortcuts only remove zero contributions), high-order
    convergence, and the three orientations agree (one_dimensional_advection_schemes.jl:108-118)"""
    def run(n, axis, flat):
        size, topo = [4, 4, 4], [0, 0, 0]
        if flat:
            size, topo = [1, 1, 1], [FLAT, FLAT, FLAT]
            topo[axis] = 0
        size[axis] = n
        ext = [(0.0, 4.0 / n)] * 3
        ext[axis] = (0.0, 1.0)
        g = oracle.Grid(tuple(size), topology=tuple(topo), x=ext[0], y=ext[1], z=ext[2])
        h = 1.0 / n
        s = (np.arange(n) + 0.5) * h
        shp = [1, 1, 1]
        shp[axis] = n
        c0 = np.exp(-((s - 0.5) ** 2) / 0.01).reshape(shp) * np.ones(size)
        m = oracle.Model(g, 1)
        vel = {"u": 0.0, "v": 0.0, "w": 0.0}
        vel["uvw"[axis]] = 1.0
        m.set(u=vel["u"] + 0 * c0, v=vel["v"] + 0 * c0, w=vel["w"] + 0 * c0, c0=c0)
        dt, nsteps = 0.01 * h, 10
        for _ in range(nsteps):
            m.time_step(dt)
        exact = np.exp(-((s - 0.5 - dt * nsteps) ** 2) / 0.01).reshape(shp) * np.ones(size)
        return np.abs(g.interior_cells(m.field("c0")) - exact).max()
    e3 = [run(n, 0, False) for n in (32, 64)]
    for axis in range(3):
        ef = [run(n, axis, True) for n in (32, 64)]
        assert np.allclose(ef, e3, rtol=1e-9), (axis, ef, e3)
    assert np.log2(e3[0] / e3[1]) > 3.5


@pytest.mark.parametrize("flat_dim", [0, 1, 2])
def test_flat_direction_equals_uniform_periodic_direction(oracle, flat_dim):
    """a two-dimensional model (one Flat direction) reproduces the three-dimensional periodic model whose fields are uniform
    along that direction: identical bits for every field whose stencils never interpolate along it, round-off for the others"""
    N = [12, 10, 8]

    def init(name, x, y, z):
        tp = 2 * np.pi
        c = [x, y, z]
        a, b = [c[d] for d in range(3) if d != flat_dim]
        f = {"u": 0.5 * np.sin(tp * a) * np.cos(tp * b) + 0.1, "v": 0.3 * np.cos(tp * a) * np.sin(tp * b) + 0.2,
             "w": -0.4 * np.cos(tp * a) * np.cos(tp * b) + 0.05, "c0": np.exp(-((a - 0.5) ** 2 + (b - 0.5) ** 2) / 0.02),
             "c1": 1.0 + 0.3 * np.sin(tp * a) * np.sin(tp * b)}
        return f[name]

    def run(flat):
        size, topo = list(N), [0, 0, 0]
        size[flat_dim] = 1 if flat else 4
        if flat:
            topo[flat_dim] = FLAT
        g = oracle.Grid(tuple(size), topology=tuple(topo), x=(0.0, 1.0), y=(0.0, 1.0), z=(0.0, 1.0))
        m = oracle.Model(g, 2)
        vals = {}
        for n in m.names():
            loc = m.loc(n)
            ax = []
            for d in range(3):
                a = (np.arange(size[d]) + (0.0 if loc[d] else 0.5)) / size[d]
                shp = [1, 1, 1]
                shp[d] = size[d]
                ax.append(a.reshape(shp))
            vals[n] = init(n, *ax) + np.zeros(size)
        m.set(**vals)
        for _ in range(5):
            m.time_step(0.1 / 12 / 0.6)
        take = [slice(None)] * 3
        take[flat_dim] = 0
        return {n: g.interior(m.field(n), m.loc(n) if n != "p" else (0, 0, 0))[tuple(take)].copy() for n in m.names() + ["p"]}, \
            m.max_abs_divergence()
    (a, da), (b, db) = run(True), run(False)
    assert da < 1e-13 and db < 1e-13
    for n in a:
        assert rel_err_(a[n], b[n]) < 1e-14, (n, rel_err_(a[n], b[n]))


def rel_err_(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# ---------------------------------------------------------------------------------------------------------------------
# closure = ScalarDiffusivity(ν, κ): the reference's data-free dynamics tests (test/test_dynamics.jl)
# ---------------------------------------------------------------------------------------------------------------------
def test_scalar_diffusivity_leaves_uniform_fields_alone(oracle):
    """test_diffusion_simple (test/test_dynamics.jl:17-32): ν = κ = 1, Δt = 1, 10 steps, a field equal to π stays π"""
    for name in ("u", "v", "c0"):
        g = oracle.Grid((4, 4, 16), topology=(P, P, B), z=(-1.0, 0.0))
        m = oracle.Model(g, 1)
        m.set_closure(nu=1.0, kappa=1.0)
        vals = {n: 0.0 for n in m.names()}
        vals[name] = np.pi
        m.set(enforce_incompressibility=False, **vals)
        for _ in range(10):
            m.time_step(1.0)
        assert np.allclose(g.interior(m.field(name), m.loc(name)), np.pi, rtol=1e-12, atol=0)


@pytest.mark.parametrize("name", ["u", "v", "c0"])
def test_scalar_diffusivity_diffusing_cosine(oracle, name):
    """test_diffusion_cosine (test/test_dynamics.jl:65-87): cos(2 z) decays as exp(-κ 4 t); κ = ν = 1, Δt = 1e-6 Lz², 5 steps,
    isapprox(atol = 1e-6, rtol = 1e-6)"""
    N, Lz = 128, np.pi / 2
    g = oracle.Grid((4, 4, N), topology=(P, P, B), z=(0.0, Lz))
    m = oracle.Model(g, 1)
    m.set_closure(nu=1.0, kappa=1.0)
    z = ((np.arange(N) + 0.5) * Lz / N).reshape(1, 1, N)
    f0 = np.cos(2 * z) + np.zeros((4, 4, N))
    vals = {n: np.zeros(g.interior(m.field(n), m.loc(n)).shape) for n in m.names()}
    vals[name] = f0
    m.set(enforce_incompressibility=False, **vals)
    dt = 1e-6 * Lz ** 2
    for _ in range(5):
        m.time_step(dt)
    exact = np.exp(-4 * m.time) * f0
    assert np.allclose(g.interior(m.field(name), m.loc(name)), exact, rtol=1e-6, atol=1e-6)
    assert np.abs(g.interior(m.field(name), m.loc(name)) - f0).max() > 1e-6          # it did diffuse


def test_viscous_taylor_green_vortex(oracle):
    """taylor_green_vortex_test (test/test_dynamics.jl:216-261): u = -sin(2πy) e^{-4π²νt}, v = sin(2πx) e^{-4π²νt}, ν = 1, N = 64,
    Δt = Δx² / (10π ν), 10 steps, max relative error < 5e-6"""
    N = 64
    g = oracle.Grid((N, N, 4))
    m = oracle.Model(g, 0)
    m.set_closure(nu=1.0)
    h = 1.0 / N
    xC, xF = ((np.arange(N) + 0.5) * h).reshape(N, 1, 1), (np.arange(N) * h).reshape(N, 1, 1)
    yC, yF = xC.reshape(1, N, 1), xF.reshape(1, N, 1)
    one = np.ones((N, N, 4))
    m.set(u=-np.sin(2 * np.pi * yC) * one, v=np.sin(2 * np.pi * xC) * one, w=0 * one)
    dt = (1 / (10 * np.pi)) * h ** 2
    for _ in range(10):
        m.time_step(dt)
    decay = np.exp(-4 * np.pi ** 2 * m.time)
    u, v = g.interior(m.field("u"), (1, 0, 0)), g.interior(m.field("v"), (0, 1, 0))
    ue, ve = -np.sin(2 * np.pi * yC) * decay * one, np.sin(2 * np.pi * xC) * decay * one
    assert np.abs((u - ue) / ue).max() < 5e-6 and np.abs((v - ve) / ve).max() < 5e-6
    assert abs(decay - 1) > 1e-3                                                        # the vortex did decay


def test_scalar_diffusivity_conserves_the_mean(oracle):
    """test_ScalarDiffusivity_budget (test/test_dynamics.jl:34-56): random field, ν = κ = 1, Δt = 1e-4 Δz², 10 steps, mean kept"""
    rng = np.random.default_rng(2)
    for name in ("u", "c0"):
        g = oracle.Grid((6, 5, 8), topology=(P, B, B), z=(-1.0, 0.0))
        m = oracle.Model(g, 1)
        m.set_closure(nu=1.0, kappa=1.0)
        vals = {n: np.zeros(g.interior(m.field(n), m.loc(n)).shape) for n in m.names()}
        vals[name] = rng.random(vals[name].shape)
        m.set(enforce_incompressibility=False, **vals)
        before = g.interior(m.field(name), m.loc(name)).mean()
        for _ in range(10):
            m.time_step(1e-4 * (1 / 8) ** 2)
        assert abs(g.interior(m.field(name), m.loc(name)).mean() - before) < 1e-13


def test_quasi_adams_bashforth_2_time_stepper(oracle):
    """QuasiAdamsBashforth2TimeStepper (TimeSteppers/quasi_adams_bashforth_2.jl:74-175): first step (and any step after Δt
    changed) is forward Euler; incompressibility after 1 and 10 steps like test/test_time_stepping.jl:124-160; second order in
    time on the diffusing cosine (constant Δt)"""
    g = oracle.Grid((16, 16, 16))
    m = _random_model(oracle, g, seed=3)
    m.time_step_ab2(1e-3)
    assert m.iteration == 1 and m.time == 1e-3 and m.max_abs_divergence() < 5e-8
    for _ in range(9):
        m.time_step_ab2(1e-3)
    assert m.iteration == 10 and m.max_abs_divergence() < 5e-8
    # Euler first step: u1 = u0 + Δt G0 (no pressure gradient for a pure-diffusion, zero-velocity problem => exact check on c)
    N, Lz = 64, np.pi / 2

    def decay_error(dt, nsteps):
        gz = oracle.Grid((4, 4, N), topology=(P, P, B), z=(0.0, Lz))
        mm = oracle.Model(gz, 1)
        mm.set_closure(nu=0.0, kappa=1.0)
        z = ((np.arange(N) + 0.5) * Lz / N).reshape(1, 1, N)
        c0 = np.cos(2 * z) + np.zeros((4, 4, N))
        mm.set(enforce_incompressibility=False, u=0 * c0, v=0 * c0, w=np.zeros((4, 4, N + 1)), c0=c0)
        lam = (2 * np.sin(2 * (Lz / N) / 2) / (Lz / N)) ** 2          # eigenvalue of the discrete operator for cos(2z)
        mm.time_step_ab2(dt)
        first = gz.interior_cells(mm.field("c0")) / c0
        assert np.allclose(first, 1 - dt * lam, rtol=1e-12)            # forward Euler
        for _ in range(nsteps - 1):
            mm.time_step_ab2(dt)
        return np.abs(gz.interior_cells(mm.field("c0")) / c0 - np.exp(-lam * dt * nsteps)).max()
    e1, e2 = decay_error(2e-4, 50), decay_error(1e-4, 100)
    assert e1 / e2 > 3.0 and e2 < 5e-6, (e1, e2)       # better than first order at these step sizes (χ = 0.1 keeps the O(χ Δt) term small)


# ---------------------------------------------------------------------------------------------------------------------
# buoyancy: BuoyancyTracer / linear SeawaterBuoyancy with the separated hydrostatic pressure anomaly
# ---------------------------------------------------------------------------------------------------------------------
def test_stratified_fluid_remains_at_rest(oracle):
    """stratified_fluid_remains_at_rest_* (test/test_dynamics.jl:263-306) with vertical gravity: b = N² z with Gradient
    conditions N² at top and bottom, Δt = 10 minutes for one hour: ∂z b stays N² everywhere and nothing moves"""
    N, L, N2 = 16, 2000.0, 1e-5
    g = oracle.Grid((4, N, N), topology=(P, B, B), x=(0.0, L), y=(0.0, L), z=(-L, 0.0))
    m = oracle.Model(g, 1)
    m.set_buoyancy_tracer(0)
    m.set_bc("c0", "bottom", "gradient", N2)
    m.set_bc("c0", "top", "gradient", N2)
    z = (-L + (np.arange(N) + 0.5) * L / N).reshape(1, 1, N)
    m.set(u=np.zeros((4, N, N)), v=np.zeros((4, N + 1, N)), w=np.zeros((4, N, N + 1)), c0=N2 * z + np.zeros((4, N, N)))
    for _ in range(6):
        m.time_step(600.0)
    b = g.interior_cells(m.field("c0"))
    assert np.allclose(np.diff(b, axis=2) / (L / N), N2, rtol=1e-12)
    assert all(np.abs(m.field(n)).max() == 0.0 for n in "uvw")
    # the hydrostatic pressure anomaly balances the stratification: ∂z pHY′ = b at the faces between cells
    p = g.interior_cells(m.field("pHY"))
    assert np.allclose(np.diff(p, axis=2) / (L / N), 0.5 * (b[:, :, 1:] + b[:, :, :-1]), rtol=1e-12)


@pytest.mark.parametrize("f,ytopo", [(0.2, FLAT), (0.2, P), (0.0, FLAT)])
def test_internal_wave_dynamics(oracle, f, ytopo):
    """internal_wave_dynamics_test (test/test_internal_wave_dynamics.jl:4-86, test/test_dynamics.jl:640-700) AS THE REFERENCE RUNS IT:
    (Periodic, Flat | Periodic, Bounded) 128 x 128 grid, BuoyancyTracer, FPlane(f = 0.2), ScalarDiffusivity(ν = κ = 1e-9), 10 steps
    of Δt = 0.01 / σ: the mean-square relative error of u stays below the reference's 1e-4 (f = 0: the pure gravity wave)"""
    Lx, Nx, Nz = 2 * np.pi, 128, 128
    Ny = 1 if ytopo == FLAT else 4
    g = oracle.Grid((Nx, Ny, Nz), topology=(P, ytopo, B), x=(0.0, Lx), y=(0.0, Lx), z=(-Lx, 0.0))
    m = oracle.Model(g, 1)
    m.set_buoyancy_tracer(0)
    m.set_closure(nu=1e-9, kappa=1e-9)
    if f:
        m.set_coriolis(f)
    z0, d, a0, mz, kx, NN = -Lx / 3, Lx / 20, 1e-3, 16, 1, 1.0
    sig = np.sqrt((NN ** 2 * kx ** 2 + f ** 2 * mz ** 2) / (kx ** 2 + mz ** 2))
    dt = 0.01 / sig
    cg = mz * sig / (kx ** 2 + mz ** 2) * (f ** 2 / sig ** 2 - 1)
    U, V = a0 * kx * sig / (sig ** 2 - f ** 2), a0 * kx * f / (sig ** 2 - f ** 2)
    W, Bm = a0 * mz * sig / (sig ** 2 - NN ** 2), a0 * mz * NN ** 2 / (sig ** 2 - NN ** 2)

    def env(z, t):
        return np.exp(-(z - cg * t - z0) ** 2 / (2 * d) ** 2)
    xc, xf = ((np.arange(Nx) + 0.5) * Lx / Nx).reshape(Nx, 1, 1), (np.arange(Nx) * Lx / Nx).reshape(Nx, 1, 1)
    zc, zf = (-Lx + (np.arange(Nz) + 0.5) * Lx / Nz).reshape(1, 1, Nz), (-Lx + np.arange(Nz + 1) * Lx / Nz).reshape(1, 1, Nz + 1)

    one, onew = np.ones((Nx, Ny, Nz)), np.ones((Nx, Ny, Nz + 1))

    def u_exact(t):
        return env(zc, t) * U * np.cos(kx * xf + mz * zc - sig * t) * one
    m.set(u=u_exact(0.0), v=env(zc, 0) * V * np.sin(kx * xc + mz * zc) * one, w=env(zf, 0) * W * np.cos(kx * xc + mz * zf) * onew,
          c0=(env(zc, 0) * Bm * np.sin(kx * xc + mz * zc) + NN ** 2 * zc) * one)
    for _ in range(10):
        m.time_step(dt)
    u = g.interior(m.field("u"), (1, 0, 0))
    ue = u_exact(m.time)
    assert np.mean((u - ue) ** 2) / np.mean(ue ** 2) < 1e-4
    assert np.mean((u - u_exact(0.0)) ** 2) / np.mean(ue ** 2) > 1e-4        # the wave did propagate


def test_passive_tracer_advection_as_the_reference_runs_it(oracle):
    """passive_tracer_advection_test (test/test_dynamics.jl:177-208): Gaussian of width L/15 carried by (U, V) = (0.5, 0.8), N = 128,
    100 steps of Δt = 0.05 (L/N) / |U|, ScalarDiffusivity(ν = κ = 1e-12), SeawaterBuoyancy, tracers (T, S): mean-square relative error
    of T below 1e-4. (Nz = 4 instead of the reference's 2: this implementation wants N >= halo in non-Flat directions.)"""
    N, L, U, V, Nt = 128, 1.0, 0.5, 0.8, 100
    d, x0, y0 = L / 15, L / 2, L / 2
    dt = 0.05 * L / N / np.sqrt(U ** 2 + V ** 2)
    g = oracle.Grid((N, N, 4), x=(0.0, L), y=(0.0, L), z=(-L, 0.0))
    m = oracle.Model(g, 2)
    m.set_closure(nu=1e-12, kappa=1e-12)
    m.set_seawater_buoyancy(0, 1)
    xc = ((np.arange(N) + 0.5) * L / N).reshape(N, 1, 1)
    yc = xc.reshape(1, N, 1)

    def T(t):
        return np.exp(-((xc - U * t - x0) ** 2 + (yc - V * t - y0) ** 2) / (2 * d ** 2)) * np.ones((N, N, 4))
    one = np.ones((N, N, 4))
    m.set(u=U * one, v=V * one, w=0 * one, c0=T(0.0), c1=0 * one)
    for _ in range(Nt):
        m.time_step(dt)
    Tn, Te = g.interior_cells(m.field("c0")), T(m.time)
    assert np.mean((Tn - Te) ** 2) / np.mean(Te ** 2) < 1e-4


def test_incompressibility_after_100_rk3_steps(oracle):
    """test/test_time_stepping.jl:124-160: 32^3, random velocities, max|div u| < 5e-8 after 100 steps too"""
    g = oracle.Grid((32, 32, 32))
    m = _random_model(oracle, g, seed=11)
    for _ in range(100):
        m.time_step(1e-4)
    assert m.iteration == 100 and m.max_abs_divergence() < 5e-8


def test_cell_advection_timescale(oracle):
    """Advection/cell_advection_timescale.jl:13-34 with the values of wall_time_step_wizard_tests (test/test_simulations.jl:14-75):
    one moving cell of u gives Δx / u₀; at rest the time-scale is infinite; on a stretched grid w uses Δzᵃᵃᶠ"""
    g = oracle.Grid((4, 4, 4))
    m = oracle.Model(g, 0)
    z = np.zeros((4, 4, 4))
    m.set(u=z, v=z, w=z, enforce_incompressibility=False)
    assert m.cell_advection_timescale() == np.inf
    u = z.copy(); u[1, 2, 3] = -7.0
    m.set(u=u, v=z, w=z, enforce_incompressibility=False)
    assert m.cell_advection_timescale() == 0.25 / 7.0
    zf = np.array([0.0, 0.1, 0.3, 0.6, 1.0])
    g = oracle.Grid((4, 4, 4), topology=(oracle.PERIODIC, oracle.PERIODIC, oracle.BOUNDED), z=zf)
    m = oracle.Model(g, 0)
    w = np.zeros((4, 4, 5)); w[0, 0, 2] = 2.0        # face k = 3 (1-based): Δzᵃᵃᶠ = zc[3] - zc[2] = 0.45 - 0.2
    m.set(u=z, v=z, w=w, enforce_incompressibility=False)
    assert np.isclose(m.cell_advection_timescale(), 0.25 / 2.0, rtol=1e-15)


# ----------------------------------------------------------------------------------------------------------------------
# AnisotropicMinimumDissipation. The reference's tests only time-step a model with it (test_time_stepping.jl:257,
# 400), so the restatement is pinned on flows whose eddy coefficients follow from the closure's definition
# (anisotropic_minimum_dissipation.jl:152-196): νₑ = max(0, -Cν δ² r / q), κₑ = max(0, -Cκ δ² ϑ / σ), δ² = 3 / Σ 1/(2Δ)².
# ----------------------------------------------------------------------------------------------------------------------
def _linear_fields(oracle, g, a, b, c, grad_c):
    """parent arrays (halos included) of u = a x, v = b y, w = c z and of a tracer with constant gradient grad_c"""
    out = []
    for loc, fn in (((1, 0, 0), lambda x, y, z: a * x + 0 * y + 0 * z), ((0, 1, 0), lambda x, y, z: b * y + 0 * x + 0 * z),
                    ((0, 0, 1), lambda x, y, z: c * z + 0 * x + 0 * y),
                    ((0, 0, 0), lambda x, y, z: grad_c[0] * x + grad_c[1] * y + grad_c[2] * z)):
        shape = g.parent_size(loc)
        ax = []
        for d in range(3):
            n = shape[d]
            d0 = g.dc[d][0]                                 # regular spacing in these tests
            idx = np.arange(n) - g.H[d]
            coord = (idx + (0.0 if loc[d] else 0.5)) * d0
            s = [1, 1, 1]; s[d] = n
            ax.append(coord.reshape(s))
        out.append(np.asfortranarray(fn(*ax) * np.ones(shape)))
    return out


@pytest.mark.parametrize("spacing", [(0.25, 0.25, 0.25), (0.5, 0.25, 0.125)])
def test_amd_known_answers(oracle, spacing):
    N = (6, 6, 6)
    topo = (oracle.BOUNDED,) * 3
    g = oracle.Grid(N, topology=topo, x=(0.0, N[0] * spacing[0]), y=(0.0, N[1] * spacing[1]), z=(0.0, N[2] * spacing[2]))
    delta2 = 3.0 / sum(1.0 / (2 * d) ** 2 for d in spacing)
    inner = (slice(4, -4),) * 3
    # 1. axisymmetric strain u = (x, y, -2z): r = a³ + b³ + c³ = -6, q = 6  =>  νₑ = Cν δ²; tracer c = z: ϑ/σ = c  =>  κₑ = 2 Cκ δ²
    u, v, w, c = _linear_fields(oracle, g, 1.0, 1.0, -2.0, (0.0, 0.0, 1.0))
    nu, (kap,) = oracle.compute_amd_diffusivities(g, 1 / 3, [1 / 12], u, v, w, [c])
    assert np.allclose(nu[inner], delta2 / 3, rtol=1e-13, atol=0)
    assert np.allclose(kap[inner], 2 * delta2 / 12, rtol=1e-13, atol=0)
    # the same strain, tracer gradient along x: ϑ/σ = a = 1 > 0  =>  κₑ clipped to zero
    _, _, _, cx = _linear_fields(oracle, g, 1.0, 1.0, -2.0, (1.0, 0.0, 0.0))
    _, (kap,) = oracle.compute_amd_diffusivities(g, 1 / 3, [1 / 3], u, v, w, [cx])
    assert np.all(kap[inner] == 0.0)
    # 2. reversed strain (-x, -y, 2z): r = +6  =>  clipped to zero (no backscatter)
    u2, v2, w2, _ = _linear_fields(oracle, g, -1.0, -1.0, 2.0, (0.0, 0.0, 1.0))
    nu, _ = oracle.compute_amd_diffusivities(g, 1 / 3, [], u2, v2, w2, [])
    assert np.all(nu[inner] == 0.0)
    # 3. laminar shear u = S z: the minimum-dissipation property -- no eddy viscosity
    shape = g.parent_size((1, 0, 0))
    zc = ((np.arange(shape[2]) - g.H[2] + 0.5) * spacing[2]).reshape(1, 1, -1)
    ush = np.asfortranarray(3.0 * zc * np.ones(shape))
    nu, _ = oracle.compute_amd_diffusivities(g, 1 / 3, [], ush, 0 * v, 0 * w, [])
    assert np.all(nu[inner] == 0.0)
    # 5. symmetric gradient tensor with off-diagonal terms, u = (x + a y, a x + y, -2z) on isotropic cells: r = tr S³ = 6a² - 6,
    #    q = 6 + 2a²  =>  νₑ = Cν δ² (6 - 6a²) / (6 + 2a²)   (exercises the interpolated ffc terms)
    if spacing[0] == spacing[1] == spacing[2]:
        a = 0.5
        shp_u, shp_v = g.parent_size((1, 0, 0)), g.parent_size((0, 1, 0))
        xs = lambda n, face: ((np.arange(n) - 3 + (0.0 if face else 0.5)) * spacing[0])          # noqa: E731
        uo = np.asfortranarray(xs(shp_u[0], 1).reshape(-1, 1, 1) + a * xs(shp_u[1], 0).reshape(1, -1, 1) + np.zeros(shp_u))
        vo = np.asfortranarray(a * xs(shp_v[0], 0).reshape(-1, 1, 1) + xs(shp_v[1], 1).reshape(1, -1, 1) + np.zeros(shp_v))
        nu, _ = oracle.compute_amd_diffusivities(g, 1 / 3, [], uo, vo, w, [])
        assert np.allclose(nu[inner], (delta2 / 3) * (6 - 6 * a * a) / (6 + 2 * a * a), rtol=1e-13, atol=0)
    # 4. fluid at rest: q = 0 and σ = 0 short-circuits
    nu, (kap,) = oracle.compute_amd_diffusivities(g, 1 / 3, [1 / 3], 0 * u, 0 * v, 0 * w, [0 * c])
    assert np.all(nu == 0.0) and np.all(kap == 0.0)


def test_amd_model_time_steps_and_dissipates(oracle):
    """time_stepping_works_with_closure (test_time_stepping.jl:41-57, 400) + what an LES closure is for: with the eddy viscosity
    a random flow loses kinetic energy faster than without, and νₑ, κₑ are nonnegative with filled halos"""
    g = oracle.Grid((12, 12, 12), topology=(oracle.PERIODIC, oracle.PERIODIC, oracle.BOUNDED))
    rng = np.random.default_rng(5)
    init = {n: rng.standard_normal(s) for n, s in (("u", (12, 12, 12)), ("v", (12, 12, 12)), ("w", (12, 12, 13)), ("c0", (12, 12, 12)))}
    init["w"][:, :, 0] = init["w"][:, :, -1] = 0.0
    ke = []
    for amd in (False, True):
        m = oracle.Model(g, 1)
        if amd:
            m.set_amd(C=1 / 3)
        m.set(**init)
        for _ in range(5):
            m.time_step(2e-3)
        assert m.iteration == 5 and all(np.isfinite(m.field(n)).all() for n in ("u", "v", "w", "c0"))
        ke.append(sum(float((g.interior(m.field(n), m.loc(n)) ** 2).sum()) for n in ("u", "v", "w")))
        if amd:
            nu, ka = m.field("nu_e"), m.field("kappa_e0")
            assert nu.min() >= 0 and ka.min() >= 0 and nu.max() > 0 and ka.max() > 0
            assert np.array_equal(nu[:3], nu[-6:-3]) and np.array_equal(nu[:, :, 2], nu[:, :, 3])     # periodic x; zero-gradient bottom
    assert ke[1] < ke[0]


def test_linear_field_dependent_flux_condition(oracle):
    """FluxBoundaryCondition(Jˢ, field_dependencies = :S, parameters = rate) with Jˢ(x, y, t, S, rate) = -rate S, the evaporation
    condition of examples/ocean_wind_mixing_and_convection.jl:125-136, as the linear family a + b φ: for a fluid at rest the top cell
    obeys dS/dt = (rate / Δz) S, which RK3 integrates as the third-order Taylor polynomial; nothing else changes"""
    g = oracle.Grid((4, 4, 8), topology=(oracle.PERIODIC, oracle.PERIODIC, oracle.BOUNDED), z=(-2.0, 0.0))
    m = oracle.Model(g, 2)
    rate = 0.3
    m.set_linear_flux_bc("c1", "top", 0.0, -rate, "c1")
    z = np.zeros((4, 4, 8))
    S0 = 35.0 + z
    m.set(u=z, v=z, w=np.zeros((4, 4, 9)), c0=20.0 + z, c1=S0)
    dt = 0.05
    m.time_step(dt)
    S = g.interior_cells(m.field("c1"))
    x = rate * dt / 0.25
    assert np.allclose(S[:, :, -1], 35.0 * (1 + x + x * x / 2 + x ** 3 / 6), rtol=1e-14, atol=0)
    assert np.array_equal(S[:, :, :-1], S0[:, :, :-1]) and np.array_equal(g.interior_cells(m.field("c0")), 20.0 + z)
    # validation: Bounded side, Center along it, dependency at the same tangential location
    with pytest.raises(ValueError):
        m.set_linear_flux_bc("c1", "east", 0.0, 1.0, "c1")
    with pytest.raises(ValueError):
        m.set_linear_flux_bc("c1", "top", 0.0, 1.0, "u")
