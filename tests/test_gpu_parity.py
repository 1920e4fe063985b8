"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances: halo / index work is bit-exact (`==`, like test/test_halo_regions.jl:22-41). Floating-point fields: 1e-12
relative (BASELINE.json north_star) after 10 RK3 steps; single kernels are expected to be bit-identical to the oracle
because both sides evaluate the same IEEE operation sequence (FMA only at the reference's @muladd sites)."""
import numpy as np
import pytest

from helpers import field_pairs, make_pair, rel_err, set_both, smooth_state, tanh_faces

pytestmark = pytest.mark.gpu

TOPOS = [("Periodic", "Periodic", "Periodic"), ("Periodic", "Periodic", "Bounded")]
# topologies with Bounded x / y: cosine transforms in the pressure solver, wall fallbacks of the advection scheme in x / y
# (the reference's list, test/test_poisson_solvers.jl:58-98)
TOPOS_XY = [("Periodic", "Bounded", "Bounded"), ("Bounded", "Bounded", "Bounded"), ("Bounded", "Periodic", "Periodic"),
            ("Periodic", "Bounded", "Periodic"), ("Bounded", "Periodic", "Bounded")]


@pytest.mark.parametrize("size", [(8, 8, 8), (16, 9, 5), (3, 3, 3)])
@pytest.mark.parametrize("topology", TOPOS + TOPOS_XY)
def test_halo_fill_bit_exact(ocn, oracle, arch, size, topology):
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology)
    rng = np.random.default_rng(7)
    for name, f in m_gpu.fields().items():
        a = rng.standard_normal(f.shape)            # random data EVERYWHERE, halos included
        f.set_parent(a)
        b = np.asfortranarray(a.copy())
        loc = tuple(1 if l is ocn.Face else 0 for l in f.loc)
        for fill_open in (False, True):
            ocn.fill_halo_regions(f, fill_open_bcs=fill_open)
            g_cpu.fill_halo_regions(b, loc, fill_open)
            assert np.array_equal(f.parent(), b), (name, fill_open)


@pytest.mark.parametrize("topology", TOPOS + TOPOS_XY)
def test_tendencies_match_oracle(ocn, oracle, arch, topology):
    size = (16, 12, 10)
    z = tanh_faces(size[2]) if topology[2] == "Bounded" else None
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=z)
    # no projection: the FFTs differ at round-off between rocFFT and the oracle, tendencies need IDENTICAL inputs
    set_both(ocn, m_gpu, m_cpu, seed=11, enforce_incompressibility=False)
    ocn.update_state(m_gpu, True)
    m_cpu.update_state(True)
    # per-field kernels (the reference's launch structure), all-fields register-window kernel, the one-field-per-workgroup kernel (default)
    for impl, lds in ((0, 0), (1, 0), (2, 0)):
        m_gpu.set_option("tendency_impl", impl)
        for n in m_gpu.fields():
            m_gpu.tendency(n).set_parent(np.zeros(m_gpu.tendency(n).shape))
        ocn.update_state(m_gpu, True)
        for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
            G_gpu = m_gpu.tendency(n).parent()
            G_cpu = m_cpu.field("G" + cn)
            assert np.array_equal(G_gpu, G_cpu), (impl, lds, n, np.abs(G_gpu - G_cpu).max())
    m_gpu.set_option("tendency_impl", 2)


@pytest.mark.parametrize("topology,stretched", [(TOPOS[0], False), (TOPOS[1], True), (TOPOS[1], False),
                                                (TOPOS_XY[0], False), (TOPOS_XY[1], False), (TOPOS_XY[1], True),
                                                (TOPOS_XY[2], False), (TOPOS_XY[3], False), (TOPOS_XY[4], True)])
def test_time_step_parity_10_steps(ocn, oracle, arch, topology, stretched):
    size = (16, 16, 16)
    z = tanh_faces(size[2]) if stretched else None
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=z)
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.1 * g_gpu.Δxᶜᵃᵃ / 0.6
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    assert m_gpu.clock.iteration == 10 and m_cpu.iteration == 10
    assert m_gpu.clock.time == m_cpu.time
    for name, a, b in field_pairs(m_gpu, m_cpu):
        ia = a[3:-3, 3:-3, 3:-3]
        ib = b[3:-3, 3:-3, 3:-3]
        assert rel_err(ia, ib) < 1e-12, (name, rel_err(ia, ib))
    assert ocn.max_abs_divergence(m_gpu) < 5e-8      # test/test_time_stepping.jl:124-160


def test_poisson_fft_solver_residual(ocn, oracle, arch):
    """∇²ϕ ≈ R for random R (test/dependencies_for_poisson_solvers.jl:111-129), PPP, sizes of test_poisson_solvers.jl:62-89"""
    rng = np.random.default_rng(3)
    for size in [(16, 16, 16), (11, 16, 7), (8, 13, 27)]:
        grid = ocn.RectilinearGrid(arch, size=size, extent=(1, 1, 1))
        solver = ocn.FFTBasedPoissonSolver(grid)
        R = rng.standard_normal(size)
        R -= R.mean()
        solver.set_source_term(R)
        phi = ocn.CenterField(grid)
        ocn.solve(phi, solver)
        ocn.fill_halo_regions(phi)
        p = phi.parent()
        dx, dy, dz = grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, grid.Δzᵃᵃᶜ[0]
        c = p[3:-3, 3:-3, 3:-3]
        lap = ((p[4:-2, 3:-3, 3:-3] - 2 * c + p[2:-4, 3:-3, 3:-3]) / dx**2 +
               (p[3:-3, 4:-2, 3:-3] - 2 * c + p[3:-3, 2:-4, 3:-3]) / dy**2 +
               (p[3:-3, 3:-3, 4:-2] - 2 * c + p[3:-3, 3:-3, 2:-4]) / dz**2)
        assert np.allclose(lap, R, atol=1e-9 * np.abs(R).max()), size
        # against the oracle's own FFT
        g_cpu = oracle.Grid(size)
        s_cpu = oracle.PoissonSolver(g_cpu, 0)
        s_cpu.rhs[...] = R
        p_cpu = g_cpu.zeros(oracle.LOC["c"])
        s_cpu.solve(p_cpu)
        assert rel_err(c, p_cpu[3:-3, 3:-3, 3:-3]) < 1e-12


@pytest.mark.parametrize("size", [(8, 8, 8), (16, 16, 16), (12, 10, 32), (4, 6, 64), (6, 5, 128), (4, 4, 512), (10, 6, 7), (8, 8, 1024), (8, 1024, 8), (16, 512, 4)])
def test_solve_for_pressure_real_transform_path(ocn, oracle, arch, size):
    """solve_for_pressure! as the model runs it (real-to-complex transforms; for Nz = 2^m in 8..512 the z transform, the spectral
    divide and the inverse z transform are one fused pass, zline_solve_kernel -- even and odd m; other Nz: rocFFT 3-D plans)
    against the oracle's source term + complex FFT solve"""
    rng = np.random.default_rng(size[2])
    grid = ocn.RectilinearGrid(arch, size=size, extent=(1, 1, 1))
    solver = ocn.FFTBasedPoissonSolver(grid)
    g_cpu = oracle.Grid(size)
    U, A = [], []
    for F, loc in ((ocn.XFaceField, "u"), (ocn.YFaceField, "v"), (ocn.ZFaceField, "w")):
        f = F(grid)
        a = np.asfortranarray(rng.standard_normal(f.shape))
        f.set_parent(a)
        ocn.fill_halo_regions(f)
        g_cpu.fill_halo_regions(a, oracle.LOC[loc])
        U.append(f)
        A.append(a)
    p = ocn.CenterField(grid)
    ocn.solve_for_pressure(p, solver, U)
    s_cpu = oracle.PoissonSolver(g_cpu, 0)
    s_cpu.rhs[...] = g_cpu.source_term(*A)
    p_cpu = g_cpu.zeros(oracle.LOC["c"])
    s_cpu.solve(p_cpu)
    assert rel_err(p.parent()[3:-3, 3:-3, 3:-3], p_cpu[3:-3, 3:-3, 3:-3]) < 1e-12
    solver.close()


@pytest.mark.parametrize("topology", TOPOS + TOPOS_XY)
@pytest.mark.parametrize("kind", ["fft", "tridiagonal"])
def test_poisson_solvers_all_topologies(ocn, oracle, arch, topology, kind):
    """∇²ϕ = ∇·U-like random source, every topology of test/test_poisson_solvers.jl:58-98 (cosine transforms on Bounded
    directions, discrete_transforms.jl:108-175), FFT-based and Fourier-tridiagonal, against the oracle's direct DCT/FFT"""
    if kind == "tridiagonal" and topology[2] != "Bounded":
        pytest.skip("FourierTridiagonalPoissonSolver needs a Bounded z")
    rng = np.random.default_rng(17)
    for size in [(16, 16, 16), (11, 16, 7), (8, 13, 27)]:
        topo_cls = tuple(getattr(ocn, t) for t in topology)
        z = tanh_faces(size[2]) if kind == "tridiagonal" else ((-1.0, 0.0) if topology[2] == "Bounded" else (0.0, 1.0))
        grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=topo_cls)
        solver = ocn.FFTBasedPoissonSolver(grid) if kind == "fft" else ocn.FourierTridiagonalPoissonSolver(grid)
        R = rng.standard_normal(size)
        R -= R.mean()
        g_cpu = oracle.Grid(size, topology=tuple(1 if t == "Bounded" else 0 for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
        s_cpu = oracle.PoissonSolver(g_cpu, 0 if kind == "fft" else 1)
        if kind == "tridiagonal":       # both sides take the Δzᶜ-weighted source (solve_for_pressure.jl:36-42); compatible: Σ = 0
            R = R * np.diff(np.asarray(z))[None, None, :]
            R -= R.mean()
        solver.set_source_term(R)
        s_cpu.rhs[...] = R
        phi = ocn.CenterField(grid)
        ocn.solve(phi, solver)
        p_cpu = g_cpu.zeros(oracle.LOC["c"])
        s_cpu.solve(p_cpu)
        got = phi.parent()[3:-3, 3:-3, 3:-3]
        assert rel_err(got, p_cpu[3:-3, 3:-3, 3:-3]) < 1e-12, (size, rel_err(got, p_cpu[3:-3, 3:-3, 3:-3]))
        solver.close()


def test_batched_tridiagonal_vs_dense(ocn, arch):
    """test/test_batched_tridiagonal_solver.jl:7-93: against a dense solve"""
    rng = np.random.default_rng(5)
    Nx, Ny, Nz = 5, 4, 16
    a = rng.random(Nz - 1)
    c = rng.random(Nz - 1)
    b = 3 + rng.random((Nx, Ny, Nz))
    f = rng.standard_normal((Nx, Ny, Nz)) + 1j * rng.standard_normal((Nx, Ny, Nz))
    phi = ocn.batched_tridiagonal_solve_z(a, b, c, f)
    for i in range(Nx):
        for j in range(Ny):
            M = np.diag(b[i, j]) + np.diag(a, -1) + np.diag(c, 1)
            assert np.allclose(phi[i, j], np.linalg.solve(M, f[i, j]), rtol=1e-12, atol=1e-13)


# ---------------------------------------------------------------------------------------------------------------------
# boundary conditions with constant values (fill_halo_regions_value_gradient.jl, fill_halo_regions_open.jl, compute_flux_bcs.jl)
# ---------------------------------------------------------------------------------------------------------------------
BC_SETS = {
    "c": dict(west=("Value", 1.5), east=("Gradient", -0.7), south=("Gradient", 0.3), north=("Value", -2.0),
              bottom=("Value", 0.25), top=("Flux", 4.0)),
    "u": dict(south=("Value", 0.0), north=("Gradient", 0.1), bottom=("Value", 0.0), top=("Flux", -1e-3), west=("Open", 0.2)),
    "w": dict(west=("Gradient", 0.5), north=("Value", 1.0), top=("Open", 0.125), bottom=("Open", -0.25)),
}


def _fbcs(ocn, spec):
    return ocn.FieldBoundaryConditions(**{s: ocn.BoundaryCondition(k, v) for s, (k, v) in spec.items()})


def _oracle_bcs(spec):
    return {s: (k.lower(), v) for s, (k, v) in spec.items()}          # v: a number or a 2-D array (array-valued condition)


@pytest.mark.parametrize("stretched", [False, True])
def test_bc_halo_fills_and_flux_tendencies_bit_exact(ocn, oracle, arch, stretched):
    size = (9, 7, 6)
    topology = ("Bounded", "Bounded", "Bounded")
    z = tanh_faces(size[2]) if stretched else None
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=z)
    rng = np.random.default_rng(21)
    flds = m_gpu.fields()
    for name, key in (("T", "c"), ("u", "u"), ("w", "w")):
        f = flds[name]
        a = rng.standard_normal(f.shape)
        f.set_parent(a)
        b = np.asfortranarray(a.copy())
        loc = tuple(1 if l is ocn.Face else 0 for l in f.loc)
        for fill_open in (False, True):
            ocn.fill_halo_regions(f, fill_open_bcs=fill_open, boundary_conditions=_fbcs(ocn, BC_SETS[key]))
            g_cpu.fill_halo_regions(b, loc, fill_open, bcs=_oracle_bcs(BC_SETS[key]))
            assert np.array_equal(f.parent(), b), (name, fill_open)
        # flux divergence of the Flux conditions on a tendency field at the same location
        if key != "w":
            G = m_gpu.tendency(name)
            ga = rng.standard_normal(G.shape)
            G.set_parent(ga)
            gb = np.asfortranarray(ga.copy())
            ocn.compute_flux_bcs(G, _fbcs(ocn, BC_SETS[key]))
            g_cpu.compute_flux_bcs(gb, loc, _oracle_bcs(BC_SETS[key]))
            assert np.array_equal(G.parent(), gb), name
    # validation mirrors the reference's: no Value condition on a wall-normal component, nothing on Periodic sides
    with pytest.raises(ocn.OcnError):
        ocn.fill_halo_regions(flds["w"], boundary_conditions=ocn.FieldBoundaryConditions(top=ocn.ValueBoundaryCondition(1.0)))


def test_model_with_boundary_conditions_matches_oracle(ocn, oracle, arch):
    """the configuration-5 style set-up: wind-stress Flux on u at the top, surface Flux + bottom Gradient on T,
    Value on S, on a stretched (Periodic, Periodic, Bounded) grid; 10 RK3 steps against the oracle"""
    size = (16, 16, 12)
    topology = ("Periodic", "Periodic", "Bounded")
    spec = {"u": dict(top=("Flux", -2e-3), bottom=("Value", 0.0)), "v": dict(bottom=("Value", 0.0)),
            "T": dict(top=("Flux", 5e-3), bottom=("Gradient", 0.4)), "S": dict(top=("Value", 35.1))}
    topo_cls = tuple(getattr(ocn, t) for t in topology)
    z = tanh_faces(size[2])
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=topo_cls)
    g_cpu = oracle.Grid(size, topology=(0, 0, 1), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, advection=ocn.WENO(), tracers=("T", "S"),
                                    boundary_conditions={n: _fbcs(ocn, s) for n, s in spec.items()})
    m_cpu = oracle.Model(g_cpu, 2)
    cname = {"u": "u", "v": "v", "w": "w", "T": "c0", "S": "c1"}
    for n, s in spec.items():
        for side, (k, v) in s.items():
            m_cpu.set_bc(cname[n], side, k.lower(), v)
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.1 * g_gpu.Δxᶜᵃᵃ / 0.6
    T0 = m_gpu.fields()["T"].parent()[3:-3, 3:-3, 3:-3].copy()
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    for name, a, b in field_pairs(m_gpu, m_cpu):
        ia, ib = a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]
        assert rel_err(ia, ib) < 1e-12, (name, rel_err(ia, ib))
    # the surface flux really acts: volume-integrated T changes by -(Q_top - 0) * area * t (bottom condition is a Gradient: no flux term)
    dz = np.diff(z)[None, None, :]
    T1 = m_gpu.fields()["T"].parent()[3:-3, 3:-3, 3:-3]
    change = ((T1 - T0) * dz).sum() / (size[0] * size[1])
    assert abs(change - (-5e-3 * 10 * dt)) < 1e-12
    with pytest.raises(ocn.OcnError):
        ocn.NonhydrostaticModel(grid=g_gpu, boundary_conditions={"T": _fbcs(ocn, dict(west=("Flux", 1.0)))})


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("topology", TOPOS)
def test_fused_substep_is_bit_identical_to_separate_kernels(ocn, arch, topology, impl):
    """RK3 substeps of stages 2 and 3 fused into the preceding tendency evaluation (second set of prognostic arrays, swapped
    twice per step) against rk3_substep! as its own launch: same IEEE operation order => identical bits, stable pointers"""
    size = (64, 20, 18)
    z = tanh_faces(size[2]) if topology[2] == "Bounded" else (0.0, 1.0)
    out = []
    for fuse in (1, 0):
        grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=tuple(getattr(ocn, t) for t in topology))
        model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
        model.set_option("tendency_impl", impl)
        model.set_option("fuse_substep", fuse)
        assert model.get_option("fuse_substep_active") == fuse and model.get_option("fused_tendency_active") == 1
        from helpers import smooth_state
        ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, seed=5))
        views = dict(model.fields())                      # Field views created BEFORE stepping must stay valid
        for _ in range(3):
            ocn.time_step(model, 0.1 * grid.Δxᶜᵃᵃ / 0.6)
        out.append({n: f.parent() for n, f in views.items()} | {"p": model.pressures.pNHS.parent(),
                                                               "Gu": model.tendency("u").parent()})
        model.close()
    for n in out[0]:
        assert np.array_equal(out[0][n], out[1][n]), n


def test_fast_reciprocals_are_correctly_rounded(ocn, arch):
    """the Float32 reciprocal inside newton_div (v_rcp_f32 + one Markstein step) is compared EXHAUSTIVELY -- all 2^23
    significands of a binade, binades spanning the range the WENO argument beta + eps can take -- with the compiler's correctly
    rounded divide; the Float64 reciprocal of the weight normalisation (the IEEE divide sequence without its scale / fixup
    instructions) on 2^28 sampled inputs over the exponents sum(alpha) can take. Zero mismatches = same bits as the reference's
    `1f0 / x` and `1 / x`."""
    import ctypes as C
    from oldoceananigans_jl_amd import _lib
    L = _lib.lib()
    bad = C.c_ulonglong()
    for exponent in (-27, -20, -8, -1, 0, 1, 7, 23, 40, 63):          # beta + eps in [1e-8, ~1e19]
        _lib.check(L.ocn_debug_rcp_check(1, exponent, C.byref(bad)))
        assert bad.value == 0, (exponent, bad.value)
    _lib.check(L.ocn_debug_rcp64_check(1 << 28, 0, 160, 12345, C.byref(bad)))
    assert bad.value == 0, bad.value
    _lib.check(L.ocn_debug_rcp64_check(1 << 24, -60, 0, 99, C.byref(bad)))       # below the range in use: still exact
    assert bad.value == 0, bad.value


# ---------------------------------------------------------------------------------------------------------------------
# Flat directions (Grids/grid_utils.jl, Operators/difference_operators.jl:30-49, Advection/flat_advective_fluxes.jl): two- and
# one-dimensional models, the set-up of the reference's own WENO convergence test
# (validation/convergence_tests/src/OneDimensionalGaussianAdvectionDiffusion.jl:14-40 uses (Nx, 1, 1) Flat grids)
# ---------------------------------------------------------------------------------------------------------------------
FLAT_CASES = [(("Periodic", "Flat", "Bounded"), (16, 1, 12), True), (("Periodic", "Flat", "Periodic"), (16, 1, 16), False),
              (("Flat", "Periodic", "Bounded"), (1, 12, 10), False), (("Bounded", "Periodic", "Flat"), (12, 16, 1), False),
              (("Periodic", "Flat", "Flat"), (32, 1, 1), False), (("Flat", "Flat", "Bounded"), (1, 1, 16), True)]


@pytest.mark.parametrize("topology,size,stretched", FLAT_CASES)
def test_flat_topologies_match_oracle(ocn, oracle, arch, topology, size, stretched):
    z = tanh_faces(size[2]) if stretched else None
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=z)
    assert m_gpu.fields()["v"].shape == g_cpu.parent_size((0, 1, 0))
    # tendencies on identical random inputs: bit-identical
    set_both(ocn, m_gpu, m_cpu, seed=3, enforce_incompressibility=False)
    ocn.update_state(m_gpu, True)
    m_cpu.update_state(True)
    for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
        assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), n
    # 10 RK3 steps from a smooth state
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.1 * min(d for d, t in zip((g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, float(np.min(g_gpu.Δzᵃᵃᶜ))), topology) if t != "Flat") / 0.6
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    H = [0 if t == "Flat" else 3 for t in topology]
    core = tuple(slice(h, -h if h else None) for h in H)
    for name, a, b in field_pairs(m_gpu, m_cpu):
        assert rel_err(a[core], b[core]) < 1e-12, (name, rel_err(a[core], b[core]))
    assert ocn.max_abs_divergence(m_gpu) < 5e-8


# ---------------------------------------------------------------------------------------------------------------------
# closure = ScalarDiffusivity(ν, κ)
# ---------------------------------------------------------------------------------------------------------------------
CLOSURE_CASES = [(("Periodic", "Periodic", "Periodic"), (16, 12, 10), False), (("Periodic", "Periodic", "Bounded"), (16, 12, 10), True),
                 (("Bounded", "Bounded", "Bounded"), (9, 8, 7), True), (("Periodic", "Flat", "Bounded"), (16, 1, 12), False)]


@pytest.mark.parametrize("topology,size,stretched", CLOSURE_CASES)
def test_scalar_diffusivity_matches_oracle(ocn, oracle, arch, topology, size, stretched):
    z = tanh_faces(size[2]) if stretched else None
    topo_cls = tuple(getattr(ocn, t) for t in topology)
    zc = z if z is not None else ((-1.0, 0.0) if topology[2] == "Bounded" else (0.0, 1.0))
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=zc, topology=topo_cls)
    g_cpu = oracle.Grid(size, topology=tuple({"Periodic": 0, "Bounded": 1, "Flat": 3}[t] for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=zc)
    closure = ocn.ScalarDiffusivity(ν=2e-3, κ={"T": 1e-3, "S": 5e-4})
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, advection=ocn.WENO(), tracers=("T", "S"), closure=closure)
    assert m_gpu.get_option("fuse_substep_active") == 1          # the substep rides in the epilogue pass that adds the closure
    m_cpu = oracle.Model(g_cpu, 2)
    m_cpu.set_closure(nu=2e-3, kappa=[1e-3, 5e-4])
    # tendencies (advection + closure) on identical random inputs: bit-identical
    set_both(ocn, m_gpu, m_cpu, seed=8, enforce_incompressibility=False)
    ocn.update_state(m_gpu, True)
    m_cpu.update_state(True)
    for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
        assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), n
    # 10 RK3 steps
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.1 * min(d for d, t in zip((g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, float(np.min(g_gpu.Δzᵃᵃᶜ))), topology) if t != "Flat") / 0.6
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    core = tuple(slice(None) if t == "Flat" else slice(3, -3) for t in topology)
    for name, a, b in field_pairs(m_gpu, m_cpu):
        assert rel_err(a[core], b[core]) < 1e-12, (name, rel_err(a[core], b[core]))


@pytest.mark.parametrize("topology,stretched", [(TOPOS[0], False), (TOPOS[1], True)])
def test_quasi_adams_bashforth_2_matches_oracle(ocn, oracle, arch, topology, stretched):
    """timestepper = :QuasiAdamsBashforth2 with a ScalarDiffusivity closure: Euler first step, AB2 steps, a
    change of Δt (Euler again), against the oracle"""
    size = (16, 16, 12)
    z = tanh_faces(size[2]) if stretched else ((-1.0, 0.0) if topology[2] == "Bounded" else (0.0, 1.0))
    topo_cls = tuple(getattr(ocn, t) for t in topology)
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=topo_cls)
    g_cpu = oracle.Grid(size, topology=tuple(1 if t == "Bounded" else 0 for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=("T", "S"), timestepper="QuasiAdamsBashforth2",
                                    closure=ocn.ScalarDiffusivity(ν=1e-3, κ=1e-3))
    m_cpu = oracle.Model(g_cpu, 2)
    m_cpu.set_closure(nu=1e-3, kappa=1e-3)
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.05 * g_gpu.Δxᶜᵃᵃ / 0.6
    for n in range(12):
        step = dt if n < 8 else 0.5 * dt
        ocn.time_step(m_gpu, step)
        m_cpu.time_step_ab2(step)
    assert m_gpu.clock.iteration == 12 and m_gpu.clock.time == m_cpu.time
    for name, a, b in field_pairs(m_gpu, m_cpu):
        assert rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]) < 1e-12, (name, rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]))
    assert ocn.max_abs_divergence(m_gpu) < 5e-8


# ---------------------------------------------------------------------------------------------------------------------
# buoyancy
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,topology,size", [("tracer", ("Periodic", "Periodic", "Bounded"), (16, 12, 12)),
                                                ("seawater", ("Periodic", "Periodic", "Bounded"), (16, 12, 12)),
                                                ("tracer", ("Bounded", "Flat", "Bounded"), (12, 1, 10)),
                                                ("seawater", ("Periodic", "Periodic", "Periodic"), (16, 16, 16))])
def test_buoyancy_matches_oracle(ocn, oracle, arch, kind, topology, size):
    """BuoyancyTracer / SeawaterBuoyancy(LinearEquationOfState) + ScalarDiffusivity: hydrostatic pressure anomaly bit-identical,
    tendencies bit-identical, 10 RK3 steps within 1e-12"""
    z = tanh_faces(size[2]) if topology[2] == "Bounded" else (0.0, 1.0)
    topo_cls = tuple(getattr(ocn, t) for t in topology)
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=topo_cls)
    g_cpu = oracle.Grid(size, topology=tuple({"Periodic": 0, "Bounded": 1, "Flat": 3}[t] for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
    tracers = ("b", "c") if kind == "tracer" else ("T", "S")
    buoyancy = ocn.BuoyancyTracer() if kind == "tracer" else ocn.SeawaterBuoyancy()
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=tracers, buoyancy=buoyancy, closure=ocn.ScalarDiffusivity(ν=1e-3, κ=1e-3))
    m_cpu = oracle.Model(g_cpu, 2)
    m_cpu.set_closure(nu=1e-3, kappa=1e-3)
    if kind == "tracer":
        m_cpu.set_buoyancy_tracer(0)
    else:
        m_cpu.set_seawater_buoyancy(0, 1)
    assert m_gpu.get_option("fuse_substep_active") == 1
    rng = np.random.default_rng(4)
    vals = {n: rng.standard_normal(g_gpu.interior_size(f.loc)) for n, f in m_gpu.fields().items()}
    ocn.set_model(m_gpu, enforce_incompressibility=False, **vals)
    m_cpu.set(enforce_incompressibility=False, **{cn: vals[gn] for cn, gn in zip(["u", "v", "w", "c0", "c1"], m_gpu.fields())})
    ocn.update_state(m_gpu, True)
    m_cpu.update_state(True)
    core = tuple(slice(None) if t == "Flat" else slice(3, -3) for t in topology)
    assert np.array_equal(m_gpu.pressures.pHY.parent()[core], m_cpu.field("pHY")[core])
    for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
        assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), n
    # 10 steps from a smooth, stably stratified state
    nodes = {n: g_gpu.nodes(f.loc) for n, f in m_gpu.fields().items()}
    st = smooth_state({"u": nodes["u"], "v": nodes["v"], "w": nodes["w"], "T": nodes[tracers[0]], "S": nodes[tracers[1]]}, 1234)
    zz = nodes[tracers[0]][2]
    wiggle = st["S"] - 35          # varies along every direction (a tracer that is uniform along one, with an offset, is
    # ill-conditioned for WENO: see tests/helpers.py)
    first = (0.5 * zz + 0.05 * wiggle) if kind == "tracer" else (20 + 5 * zz + wiggle)
    vals = {"u": st["u"], "v": st["v"], "w": st["w"], tracers[0]: first + 0 * st["T"], tracers[1]: st["S"]}
    ocn.set_model(m_gpu, **vals)
    m_cpu.set(**{cn: vals[gn] for cn, gn in zip(["u", "v", "w", "c0", "c1"], m_gpu.fields())})
    dt = 0.05 * min(d for d, t in zip((g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, float(np.min(g_gpu.Δzᵃᵃᶜ))), topology) if t != "Flat") / 0.6
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    for name, a, b in field_pairs(m_gpu, m_cpu):
        assert rel_err(a[core], b[core]) < 1e-12, (name, rel_err(a[core], b[core]))
    with pytest.raises(ValueError):
        ocn.NonhydrostaticModel(grid=g_gpu, tracers=("T",), buoyancy=ocn.SeawaterBuoyancy())


@pytest.mark.parametrize("topology,size", [(("Periodic", "Periodic", "Bounded"), (16, 12, 10)), (("Bounded", "Bounded", "Bounded"), (9, 8, 7)),
                                           (("Periodic", "Flat", "Bounded"), (16, 1, 12)), (("Bounded", "Periodic", "Periodic"), (10, 8, 8))])
def test_fplane_coriolis_matches_oracle(ocn, oracle, arch, topology, size):
    """coriolis = FPlane(f) with the active-weighted averages next to walls: tendencies bit-identical, 10 steps 1e-12"""
    z = tanh_faces(size[2]) if topology[2] == "Bounded" else (0.0, 1.0)
    topo_cls = tuple(getattr(ocn, t) for t in topology)
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=topo_cls)
    g_cpu = oracle.Grid(size, topology=tuple({"Periodic": 0, "Bounded": 1, "Flat": 3}[t] for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=("T", "S"), coriolis=ocn.FPlane(f=0.7))
    m_cpu = oracle.Model(g_cpu, 2)
    m_cpu.set_coriolis(0.7)
    set_both(ocn, m_gpu, m_cpu, seed=5, enforce_incompressibility=False)
    ocn.update_state(m_gpu, True)
    m_cpu.update_state(True)
    for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
        assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), n
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.05 * min(d for d, t in zip((g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, float(np.min(g_gpu.Δzᵃᵃᶜ))), topology) if t != "Flat") / 0.6
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    core = tuple(slice(None) if t == "Flat" else slice(3, -3) for t in topology)
    for name, a, b in field_pairs(m_gpu, m_cpu):
        assert rel_err(a[core], b[core]) < 1e-12, (name, rel_err(a[core], b[core]))
    assert ocn.FPlane(latitude=45).f == 2 * 7.292115e-5 * 0.7071067811865476          # 2 rotation_rate sind(45) (f_plane.jl:44); sind(45) is one ulp above sin(π / 4)


@pytest.mark.parametrize("topology,size", [(("Periodic", "Periodic", "Bounded"), (64, 12, 10)), (("Bounded", "Bounded", "Bounded"), (9, 8, 7))])
def test_fused_epilogue_is_bit_identical_to_separate_kernels(ocn, arch, topology, size):
    """Coriolis + hydrostatic pressure gradient + closure + next-stage substep as ONE pass (tendency_epilogue_kernel) against the
    stand-alone kernels and a separate rk3_substep!: identical bits after full time-steps"""
    z = tanh_faces(size[2])
    out = []
    for fused in (1, 0):
        grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=tuple(getattr(ocn, t) for t in topology))
        F = ocn.FieldBoundaryConditions
        bcs = {"T": F(top=ocn.FluxBoundaryCondition(5e-3), bottom=ocn.GradientBoundaryCondition(0.1)),
               "u": F(top=ocn.FluxBoundaryCondition(-1e-3), bottom=ocn.ValueBoundaryCondition(0.0))}
        model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), coriolis=ocn.FPlane(f=0.3), buoyancy=ocn.SeawaterBuoyancy(),
                                        closure=ocn.ScalarDiffusivity(ν=1e-3, κ={"T": 1e-3, "S": 0.0}), boundary_conditions=bcs)
        model.set_option("fused_epilogue", fused)
        model.set_option("fuse_substep", fused)
        assert model.get_option("fuse_substep_active") == fused
        ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, seed=9))
        for _ in range(3):
            ocn.time_step(model, 0.05 * grid.Δxᶜᵃᵃ / 0.6)
        out.append({n: f.parent() for n, f in model.fields().items()} | {"p": model.pressures.pNHS.parent(), "pHY": model.pressures.pHY.parent(),
                                                                    "Gu": model.tendency("u").parent(), "GT": model.tendency("T").parent()})
        model.close()
    for n in out[0]:
        # tendencies: cells the substep reads (the stand-alone Flux-condition kernel also touches wall faces excluded by
        # exclude_periphery, like the reference; nothing reads them)
        core = (slice(4, -3), slice(3, -3), slice(3, -3)) if n == "Gu" else (slice(None),) * 3
        assert np.array_equal(out[0][n][core], out[1][n][core]), n


def test_time_step_wizard_and_advection_timescale(ocn, oracle, arch):
    """wall_time_step_wizard_tests (test/test_simulations.jl:14-75) -- a single moving cell of u -- and cell_advection_timescale
    (Advection/cell_advection_timescale.jl:13-34) bit-identical to the oracle on random velocities over a stretched grid"""
    grid = ocn.RectilinearGrid(arch, size=(4, 4, 4), extent=(1, 1, 1))
    model = ocn.NonhydrostaticModel(grid=grid, tracers=())
    dx, CFL, u0 = grid.Δxᶜᵃᵃ, 0.45, 7.0
    u = np.zeros(grid.interior_size(model.velocities.u.loc))
    u[0, 0, 0] = u0
    model.velocities.u.set(u)
    W = ocn.TimeStepWizard
    dt = ocn.new_time_step(2.5, W(cfl=CFL, max_change=np.inf, min_change=0), model)
    assert np.isclose(dt, CFL * dx / u0, rtol=1e-15)
    assert np.isclose(ocn.new_time_step(1.0, W(cfl=CFL, max_change=np.inf, min_change=0.75), model), 0.75)
    assert np.isclose(ocn.new_time_step(dt, W(cfl=CFL, max_change=np.inf, min_change=0, min_Δt=1.99), model), 1.99)
    u[0, 0, 0] = u0 / 100
    model.velocities.u.set(u)
    assert np.isclose(ocn.new_time_step(1.0, W(cfl=CFL, max_change=1.1, min_change=0), model), 1.1)
    free = CFL * dx / (u0 / 100)                     # (the reference's grid has Δx = 1; here Δx = 1/4)
    assert np.isclose(ocn.new_time_step(1.99, W(cfl=CFL, max_change=np.inf, min_change=0, max_Δt=0.5 * free), model), 0.5 * free)
    m2 = ocn.NonhydrostaticModel(grid=grid, tracers=(), closure=ocn.ScalarDiffusivity(ν=1.0))
    assert np.isclose(ocn.new_time_step(1.0, W(cfl=np.inf, diffusive_cfl=0.45, max_change=np.inf, min_change=0), m2), 0.45 * dx ** 2 / 1.0)
    with pytest.raises(ValueError):
        W(min_change=1.5)
    # against the oracle
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, (12, 10, 9), ("Periodic", "Bounded", "Bounded"), z=tanh_faces(9))
    set_both(ocn, m_gpu, m_cpu, seed=2, enforce_incompressibility=False)       # identical inputs (no projection round-off)
    assert ocn.cell_advection_timescale(m_gpu) == m_cpu.cell_advection_timescale()
    # the raw-pointer entry (what a `launch!` specialisation of cell_advection_timescale would ccall) returns the same number
    import ctypes as C
    from oldoceananigans_jl_amd import _lib
    tau = C.c_double()
    V = m_gpu.velocities
    _lib.check(_lib.lib().ocn_cell_advection_timescale(g_gpu.handle, V.u.data, V.v.data, V.w.data, C.byref(tau)))
    assert tau.value == m_cpu.cell_advection_timescale()


def test_nan_checker(ocn, arch):
    """NaNChecker / hasnan (Diagnostics/nan_checker.jl:32-53; test/test_diagnostics.jl nan_checker_aborts_simulation): a NaN anywhere
    in parent(u) -- halo cells included -- stops the simulation, or raises when erroring"""
    from types import SimpleNamespace
    grid = ocn.RectilinearGrid(arch, size=(8, 6, 4), extent=(1, 1, 1))
    model = ocn.NonhydrostaticModel(grid=grid, tracers=())
    sim = SimpleNamespace(running=True, model=model)
    checker = ocn.default_nan_checker(model)
    checker(sim)
    assert sim.running and not ocn.hasnan(model) and not ocn.hasnan(model.velocities.w)
    a = model.velocities.u.parent()
    a[0, 0, 0] = np.nan                                  # a halo corner: parent(field) is what the reference scans
    model.velocities.u.set_parent(a)
    assert ocn.hasnan(model) and ocn.hasnan(model.velocities.u) and not ocn.hasnan(model.velocities.v)
    checker(sim)
    assert not sim.running
    checker.erroring = True
    with pytest.raises(RuntimeError, match="NaN found in field u"):
        checker(sim)


@pytest.mark.parametrize("topology,size,stretched", [(("Periodic", "Periodic", "Periodic"), (16, 12, 8), False),
                                                      (("Periodic", "Periodic", "Bounded"), (16, 16, 12), True),
                                                      (("Bounded", "Bounded", "Bounded"), (12, 10, 8), False)])
def test_anisotropic_minimum_dissipation_matches_oracle(ocn, oracle, arch, topology, size, stretched):
    """closure = AnisotropicMinimumDissipation (BASELINE.json configs[4]'s closure): eddy viscosity / diffusivities
    (interior and filled halos), the tendencies with array coefficients, fused vs separate epilogue, 10 RK3 steps. Parity is
    against the oracle's restatement, which analytic known answers pin (tests/test_oracle_kats.py::test_amd_known_answers) -- the
    reference holds no numbers for this closure."""
    z = tanh_faces(size[2]) if stretched else ((-1.0, 0.0) if topology[2] == "Bounded" else (0.0, 1.0))
    topo_cls = tuple(getattr(ocn, t) for t in topology)
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=topo_cls)
    g_cpu = oracle.Grid(size, topology=tuple({"Periodic": 0, "Bounded": 1}[t] for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
    closure = ocn.AnisotropicMinimumDissipation(C=1 / 3, Cκ={"T": 1 / 3, "S": 1 / 12})
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, advection=ocn.WENO(), tracers=("T", "S"), closure=closure)
    m_cpu = oracle.Model(g_cpu, 2)
    m_cpu.set_amd(C=1 / 3, Ckappa=[1 / 3, 1 / 12])
    set_both(ocn, m_gpu, m_cpu, seed=8, enforce_incompressibility=False)
    for fused in (1, 0):
        m_gpu.set_option("fused_epilogue", fused)
        ocn.update_state(m_gpu, True)
        m_cpu.update_state(True)
        D = m_gpu.diffusivity_fields
        assert np.array_equal(D.νₑ.parent(), m_cpu.field("nu_e")) and m_cpu.field("nu_e").max() > 0
        for t, name in enumerate(("T", "S")):
            assert np.array_equal(getattr(D.κₑ, name).parent(), m_cpu.field("kappa_e%d" % t)), name
        for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
            assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), (n, fused)
    m_gpu.set_option("fused_epilogue", 1)
    # the stand-alone operators give the same arrays
    flds = list(m_gpu.fields().values())
    nu = ocn.CenterField(g_gpu)
    ka = [ocn.CenterField(g_gpu), ocn.CenterField(g_gpu)]
    ocn.kernels.compute_amd_diffusivities(g_gpu, closure, ("T", "S"), flds, nu, ka)
    ocn.fill_halo_regions([nu] + ka)
    assert np.array_equal(nu.parent(), m_cpu.field("nu_e")) and np.array_equal(ka[1].parent(), m_cpu.field("kappa_e1"))
    # 10 RK3 steps
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.1 * min(g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, float(np.min(g_gpu.Δzᵃᵃᶜ))) / 0.6
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    for name, a, b in field_pairs(m_gpu, m_cpu):
        assert rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]) < 1e-12, (name, rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]))


@pytest.mark.parametrize("case", ["ppp", "ppb_physics", "bbb"])
def test_time_step_graph_replay_is_bit_identical(ocn, arch, case):
    """after the first step an RK3 time-step is captured into a hipGraph per (Δt, configuration) and replayed: same bits, same clock as
    issuing the launches one by one; a new Δt or a changed option re-captures"""
    ocn.own_stream()        # the distributed tests of this process may have pointed the library at torch's (default) stream
    physics = {}
    if case == "ppp":
        grid = ocn.RectilinearGrid(arch, size=(16, 12, 8), extent=(1, 1, 1))
    elif case == "bbb":
        grid = ocn.RectilinearGrid(arch, size=(12, 10, 8), extent=(1, 1, 1), topology=(ocn.Bounded,) * 3)
    else:
        grid = ocn.RectilinearGrid(arch, size=(16, 16, 12), x=(0, 1), y=(0, 1), z=tanh_faces(12), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
        F = ocn.FieldBoundaryConditions
        physics = dict(closure=ocn.AnisotropicMinimumDissipation(), buoyancy=ocn.SeawaterBuoyancy(), coriolis=ocn.FPlane(f=0.3),
                       boundary_conditions={"u": F(top=ocn.FluxBoundaryCondition(-1e-3)), "T": F(bottom=ocn.GradientBoundaryCondition(0.1))})
    models = []
    for use_graph in (0, 1):
        m = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), **physics)
        m.set_option("use_graph", use_graph)
        flds = m.fields()
        ocn.set_model(m, **smooth_state({n: grid.nodes(f.loc) for n, f in flds.items()}, 77))
        models.append(m)
    plain, graphed = models
    dts = [1e-3] * 5 + [7e-4] * 3
    for n, dt in enumerate(dts):
        for m in models:
            ocn.time_step(m, dt)
        if n == 5:
            graphed.set_option("fused_epilogue", 1)          # any option change invalidates the captured graph
    assert graphed.get_option("graph_failures") == 0 and plain.get_option("graph_captures") == 0
    assert graphed.get_option("graph_captures") == 3 and graphed.get_option("graph_replays") == len(dts) - 1 - 3
    assert (plain.clock.time, plain.clock.iteration, plain.clock.stage, plain.clock.last_Δt, plain.clock.last_stage_Δt) == \
           (graphed.clock.time, graphed.clock.iteration, graphed.clock.stage, graphed.clock.last_Δt, graphed.clock.last_stage_Δt)
    for name in plain.fields():
        assert np.array_equal(plain.fields()[name].parent(), graphed.fields()[name].parent()), name
        assert np.array_equal(plain.tendency(name).parent(), graphed.tendency(name).parent()), name
    assert np.array_equal(plain.pressures.pNHS.parent(), graphed.pressures.pNHS.parent())


def test_ocean_wind_mixing_and_convection_physics_matches_oracle(ocn, oracle, arch):
    """the physics of BASELINE.json configs[4] (examples/ocean_wind_mixing_and_convection.jl) at a size the oracle finishes in seconds:
    stretched Bounded z, AnisotropicMinimumDissipation, linear SeawaterBuoyancy(α = 2e-4, β = 8e-4), wind stress on u, heat flux and
    bottom gradient on T, and the field-dependent evaporation flux Jˢ = -rate S on S. 10 RK3 steps to 1e-12; the evaporation term is checked against a run without it."""
    size = (16, 16, 12)
    z = tanh_faces(size[2])
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    g_cpu = oracle.Grid(size, topology=(0, 0, 1), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
    F, rate = ocn.FieldBoundaryConditions, 2.5e-3
    bcs = {"u": F(top=ocn.FluxBoundaryCondition(-1e-3)),
           "T": F(top=ocn.FluxBoundaryCondition(4e-3), bottom=ocn.GradientBoundaryCondition(0.01)),
           "S": F(top=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(b=-rate), field_dependencies="S", parameters=rate))}
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=("T", "S"), closure=ocn.AnisotropicMinimumDissipation(),
                                    buoyancy=ocn.SeawaterBuoyancy(ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4)),
                                    boundary_conditions=bcs)
    m_cpu = oracle.Model(g_cpu, 2)
    m_cpu.set_amd()
    m_cpu.set_seawater_buoyancy(alpha=2e-4, beta=8e-4)
    m_cpu.set_bc("u", "top", "flux", -1e-3)
    m_cpu.set_bc("c0", "top", "flux", 4e-3)
    m_cpu.set_bc("c0", "bottom", "gradient", 0.01)
    m_cpu.set_linear_flux_bc("c1", "top", 0.0, -rate, "c1")
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.1 * min(g_gpu.Δxᶜᵃᵃ, float(np.min(g_gpu.Δzᵃᵃᶜ))) / 0.6
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    for name, a, b in field_pairs(m_gpu, m_cpu):
        assert rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]) < 1e-12, (name, rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]))
    # the evaporation term really acted: compare with a run that has no S condition
    m_ref = ocn.NonhydrostaticModel(grid=g_gpu, tracers=("T", "S"), closure=ocn.AnisotropicMinimumDissipation(),
                                    buoyancy=ocn.SeawaterBuoyancy(ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4)),
                                    boundary_conditions={k: v for k, v in bcs.items() if k != "S"})
    flds = m_ref.fields()
    ocn.set_model(m_ref, **smooth_state({n: g_gpu.nodes(f.loc) for n, f in flds.items()}, 1234))
    for _ in range(10):
        ocn.time_step(m_ref, dt)
    dS = m_gpu.tracers.S.interior()[:, :, -1] - m_ref.tracers.S.interior()[:, :, -1]
    assert dS.min() > 0 and np.allclose(dS / m_ref.tracers.S.interior()[:, :, -1], 10 * dt * rate / float(g_gpu.Δzᵃᵃᶜ[-4]), rtol=0.3)   # advection / mixing move some of it


def test_simulation_run_loop(ocn, arch):
    """run_basic_simulation_tests (test/test_simulations.jl:77-150): stop criteria, run!, reset!, stop-time alignment, the wizard as a
    callback; plus a TimeInterval callback whose actuation times the aligned time steps must hit"""
    grid = ocn.RectilinearGrid(arch, size=(4, 4, 4), extent=(1, 1, 1))
    model = ocn.NonhydrostaticModel(grid=grid, tracers=())
    sim = ocn.Simulation(model, Δt=3, stop_iteration=1)
    sim.running = True
    ocn.stop_iteration_exceeded(sim)
    assert sim.running
    ocn.run(sim)
    sim.running = True
    ocn.stop_iteration_exceeded(sim)
    assert not sim.running
    assert np.isclose(model.clock.time, sim.Δt) and model.clock.iteration == 1 and sim.run_wall_time > 0
    sim.running = True
    ocn.stop_time_exceeded(sim)
    assert sim.running
    sim.stop_time = 1e-12
    ocn.stop_time_exceeded(sim)
    assert not sim.running
    sim.running = True
    ocn.wall_time_limit_exceeded(sim)
    assert sim.running
    sim.wall_time_limit = 1e-12
    ocn.wall_time_limit_exceeded(sim)
    assert not sim.running
    # stops at stop_iteration
    ocn.reset(sim)
    assert model.clock.time == 0 and model.clock.iteration == 0 and model.clock.last_Δt == np.inf
    sim.stop_iteration = 3
    ocn.run(sim)
    assert model.clock.iteration == 3
    # stops at stop_time: the last step is shortened
    ocn.reset(sim)
    sim.stop_time = 20.20
    ocn.run(sim)
    assert np.isclose(model.clock.time, 20.20, rtol=1e-14) and model.clock.iteration == 7
    # the wizard as a callback
    ocn.reset(sim)
    sim.stop_iteration = 2
    sim.callbacks["wizard"] = ocn.Callback(ocn.TimeStepWizard(cfl=0.1), ocn.IterationInterval(1))
    ocn.run(sim)
    assert model.clock.iteration == 2
    del sim.callbacks["wizard"]
    # a callback on a TimeInterval: time steps land on its actuation times
    ocn.reset(sim)
    sim.Δt, sim.stop_time = 0.25, 2.0
    times = []
    sim.callbacks["record"] = ocn.Callback(lambda s: times.append(s.model.clock.time), ocn.TimeInterval(0.7))
    ocn.run(sim)
    assert np.allclose(times, [0.0, 0.7, 1.4], rtol=1e-14, atol=0) and np.isclose(model.clock.time, 2.0, rtol=1e-14)


@pytest.mark.parametrize("size,zb", [((20, 8, 8), False), ((36, 16, 10), True), ((64, 32, 16), False), ((10, 64, 12), True),
                                     ((16, 8, 8), False), ((32, 16, 10), True), ((128, 16, 12), True)])
def test_split_pressure_step_equals_library_plans(ocn, arch, size, zb):
    """the model's pressure step in split form (1-D x plans on 128-B-padded rows + the LDS column-FFT kernel for y + one kernel for the
    correction and p / Δt from the dense solution; option split_solve) against the 2-D library plans + separate kernels: same fields"""
    topo = (ocn.Periodic, ocn.Periodic, ocn.Bounded if zb else ocn.Periodic)
    z = tanh_faces(size[2]) if zb else (0.0, 1.0)
    outs = []
    for split in (1, 0):
        ocn.set_option("split_solve", split)
        try:
            grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=topo)
            model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
            ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 5))
            for _ in range(3):
                ocn.time_step(model, 0.1 * grid.Δxᶜᵃᵃ / 0.6)
            outs.append({n: f.parent() for n, f in model.fields().items()} | {"p": model.pressures.pNHS.parent()})
            div = ocn.max_abs_divergence(model)
            assert div < 1e-12
            del model
        finally:
            ocn.set_option("split_solve", 1)
    # error model of the pressure: two exact-arithmetic-equivalent transforms differ by round-off times the condition number of the
    # discrete Laplacian (anisotropic cells: 128 x 16 x 12 stretched has 1.8e3); velocities and tracers keep the 1e-12 bar
    dzmin = float(np.diff(np.asarray(z)).min()) if zb else 1.0 / size[2]
    cond = (4.0 * size[0] ** 2 + 4.0 * size[1] ** 2 + 4.0 / dzmin ** 2) / (2 * np.pi) ** 2
    for other in outs[1:]:
        for name in outs[0]:
            a, b = outs[0][name][3:-3, 3:-3, 3:-3], other[name][3:-3, 3:-3, 3:-3]
            tol = 1e-12 if name != "p" else max(1e-12, 4 * np.finfo(float).eps * cond)
            assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30), name


@pytest.mark.parametrize("N", [8, 9, 2, 3, 16, 27, 128])
def test_permutation_tables_on_device(ocn, oracle, arch, N):
    """the cosine-transform path's gather / scatter kernels move element i where the reference's permute_index / unpermute_index
    say (Solvers/index_permutations.jl:5-35; docstring tables for N = 8, 9), bit for bit like the oracle's restatement"""
    import ctypes as C
    from oldoceananigans_jl_amd import _lib
    tables = {(8, 0): [1, 8, 2, 7, 3, 6, 4, 5], (9, 0): [1, 9, 2, 8, 3, 7, 4, 6, 5],
              (8, 1): [1, 3, 5, 7, 8, 6, 4, 2], (9, 1): [1, 3, 5, 7, 9, 8, 6, 4, 2]}
    L = oracle.lib()
    for backward in (0, 1):
        out = (C.c_int * N)()
        _lib.check(_lib.lib().ocn_debug_permute_indices(N, backward, out))
        got = list(out)
        fn = L.oro_unpermute_index if backward else L.oro_permute_index
        assert got == [fn(i, N) for i in range(1, N + 1)]
        if (N, backward) in tables:
            assert got == tables[(N, backward)]


ADAPTED = [((4, 2, 4), ("Periodic", "Periodic", "Periodic")), ((8, 2, 6), ("Periodic", "Periodic", "Bounded")),
           ((2, 8, 6), ("Periodic", "Periodic", "Periodic")), ((6, 8, 2), ("Periodic", "Periodic", "Periodic")),
           ((8, 2, 6), ("Periodic", "Bounded", "Bounded")), ((2, 6, 2), ("Bounded", "Periodic", "Periodic"))]


@pytest.mark.parametrize("size,topology", ADAPTED)
def test_adapted_advection_order_matches_oracle(ocn, oracle, arch, size, topology):
    """NonhydrostaticModel(advection = WENO()) on a grid with two cells in a direction (adapt_advection_order.jl:18-96, expectations of
    test/test_nonhydrostatic_models.jl:72-91): that direction carries WENO(order=3) + Centered(order=2) and a halo of 2. Tendencies
    bit-identical to the oracle, 10 RK3 steps within 1e-12."""
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology)
    assert isinstance(m_gpu.advection, ocn.FluxFormAdvection)
    expected = tuple(min(3, n) for n in size)
    assert (ocn.required_halo_size_x(m_gpu.advection), ocn.required_halo_size_y(m_gpu.advection),
            ocn.required_halo_size_z(m_gpu.advection)) == expected
    assert m_gpu.grid.halo_size == expected == g_cpu.H and g_cpu.B == expected
    assert m_gpu.get_option("fused_tendency_active") == 0          # reduced-order directions take the per-field kernels
    set_both(ocn, m_gpu, m_cpu, seed=5, enforce_incompressibility=False)
    ocn.update_state(m_gpu, True)
    m_cpu.update_state(True)
    for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
        G_gpu, G_cpu = m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)
        assert np.array_equal(G_gpu, G_cpu), (n, np.abs(G_gpu - G_cpu).max())
    set_both(ocn, m_gpu, m_cpu, seed=6, smooth=True)
    dt = 0.02 * min(g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ)
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    Hx, Hy, Hz = expected
    umax = max(np.abs(m_cpu.field(n)).max() for n in ("u", "v", "w"))
    for name, a, b in field_pairs(m_gpu, m_cpu):
        ia, ib = a[Hx:-Hx, Hy:-Hy, Hz:-Hz], b[Hx:-Hx, Hy:-Hy, Hz:-Hz]
        assert np.all(np.isfinite(ia))
        if name == "pNHS":
            # on these few-cell grids the smooth state is almost divergence-free and |p| is tiny, while the round-off that reaches p is
            # that of the velocities: p = lap^-1(div u*) / dt carries eps |u| dx / dt (the pressure that would change u by itself in
            # one step) -- the error is measured against that scale
            pscale = max(np.abs(ib).max(), umax * max(g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ) / dt)
            assert np.max(np.abs(ia - ib)) < 1e-12 * pscale, (name, np.max(np.abs(ia - ib)), pscale)
            continue
        assert rel_err(ia, ib) < 1e-12, (name, rel_err(ia, ib))
    assert ocn.max_abs_divergence(m_gpu) < 5e-8


def test_one_cell_in_a_non_flat_direction_is_refused(ocn, arch):
    """N = 1 in a Periodic / Bounded direction: adapt_advection_order would give UpwindBiased(order=1) with a one-cell halo, into which
    the reference's Centered(order=4) advecting-velocity interpolation of the other directions' fluxes reads two cells. Refused with
    a message that names the remedy (a Flat direction)."""
    grid = ocn.RectilinearGrid(arch, size=(4, 1, 4), extent=(1, 1, 1))
    with pytest.raises(ocn.OcnError, match="Flat"):
        ocn.NonhydrostaticModel(grid=grid)


@pytest.mark.parametrize("size", [(8, 8, 8), (16, 9, 5), (3, 3, 3), (5, 4, 2)])
def test_one_launch_fill_on_bounded_z_with_boundary_conditions(ocn, oracle, arch, size):
    """(Periodic, Periodic, Bounded): bounded z fill + periodic y + periodic x as ONE launch (fill_periodic_xy_bounded_z_kernel) -- every
    cell of the parent array, the stale deep z halos included, equals the oracle's three ordered fills
    (boundary_condition_ordering.jl:17-46) and the three-launch path, with Value / Gradient / Flux / Open conditions on the z sides"""
    topology = ("Periodic", "Periodic", "Bounded")
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology)
    spec = {"T": dict(bottom=("Value", 0.25), top=("Gradient", -0.7)), "S": dict(top=("Flux", 4.0)),
            "u": dict(bottom=("Value", 0.0), top=("Flux", -1e-3)), "v": dict(bottom=("Gradient", 0.3)),
            "w": dict(top=("Open", 0.125), bottom=("Open", -0.25))}
    rng = np.random.default_rng(17)
    for name, f in m_gpu.fields().items():
        loc = tuple(1 if l is ocn.Face else 0 for l in f.loc)
        for fill_open in (False, True):
            a = rng.standard_normal(f.shape)            # random data EVERYWHERE, halos included
            out = []
            for fused in (1, 0):
                ocn.set_option("fused_halo", fused)
                f.set_parent(a)
                ocn.fill_halo_regions(f, fill_open_bcs=fill_open, boundary_conditions=_fbcs(ocn, spec[name]))
                out.append(f.parent())
            ocn.set_option("fused_halo", 1)
            b = np.asfortranarray(a.copy())
            g_cpu.fill_halo_regions(b, loc, fill_open, bcs=_oracle_bcs(spec[name]))
            assert np.array_equal(out[0], b), (name, fill_open)
            assert np.array_equal(out[0], out[1]), (name, fill_open)


def test_fused_substep_with_field_dependent_flux_conditions(ocn, arch):
    """the configs[4] physics keeps the RK3 substeps of stages 2 and 3 inside the one-pass epilogue also when a linear
    field-dependent Flux condition is set (it used to fall back to rk3_substep_kernel): bit-identical to the separate launches"""
    size = (64, 12, 10)
    z = tanh_faces(size[2])
    F, rate = ocn.FieldBoundaryConditions, 2.5e-3
    bcs = {"u": F(top=ocn.FluxBoundaryCondition(-1e-3)),
           "T": F(top=ocn.FluxBoundaryCondition(4e-3), bottom=ocn.GradientBoundaryCondition(0.01)),
           "S": F(top=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(a=1e-4, b=-rate), field_dependencies="S"),
                  bottom=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(b=rate), field_dependencies="T"))}
    out = []
    for fuse, epilogue in ((1, 1), (0, 1), (0, 0)):
        grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
        model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), closure=ocn.AnisotropicMinimumDissipation(),
                                        buoyancy=ocn.SeawaterBuoyancy(ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4)),
                                        boundary_conditions=bcs)
        model.set_option("fuse_substep", fuse)
        model.set_option("fused_epilogue", epilogue)
        assert model.get_option("fuse_substep_active") == fuse
        ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, seed=5))
        for _ in range(3):
            ocn.time_step(model, 0.05 * grid.Δxᶜᵃᵃ)
        out.append({n: f.parent() for n, f in model.fields().items()} | {"Gu": model.tendency("u").parent(), "GS": model.tendency("S").parent()})
        model.close()
    for other in out[1:]:
        for n in out[0]:
            assert np.array_equal(out[0][n], other[n]), n
    # and without any other physics: the Flux conditions alone route the substep through the epilogue
    grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), boundary_conditions=bcs)
    assert model.get_option("fuse_substep_active") == 1


@pytest.mark.parametrize("topology", [("Bounded", "Bounded", "Bounded"), ("Periodic", "Periodic", "Bounded")])
def test_array_valued_boundary_conditions_bit_exact(ocn, oracle, arch, topology):
    """array-valued Flux / Value / Gradient / Open conditions (getbc(condition::AbstractArray, i, j, ...) = condition[i, j],
    boundary_condition.jl:164) cross the C ABI as a device pointer per side: halo fills (multi-launch and the one-launch (P, P, B)
    kernel) and the flux divergence of the Flux sides are bit-identical to the oracle"""
    size = (9, 7, 6)
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=tanh_faces(size[2]))
    rng = np.random.default_rng(23)
    tang = {"west": (size[1], size[2]), "east": (size[1], size[2]), "south": (size[0], size[2]), "north": (size[0], size[2]),
            "bottom": (size[0], size[1]), "top": (size[0], size[1])}
    bounded_sides = [s for s, d in (("west", 0), ("east", 0), ("south", 1), ("north", 1), ("bottom", 2), ("top", 2)) if topology[d] == "Bounded"]
    kinds_c = {"west": "Value", "east": "Gradient", "south": "Gradient", "north": "Value", "bottom": "Value", "top": "Flux"}
    spec_c = {s: (kinds_c[s], rng.standard_normal(tang[s])) for s in bounded_sides}
    spec_w = {s: ("Open", rng.standard_normal(tang[s])) for s in ("bottom", "top")}
    flds = m_gpu.fields()
    for name, spec in (("T", spec_c), ("w", spec_w)):
        f = flds[name]
        loc = tuple(1 if l is ocn.Face else 0 for l in f.loc)
        a = rng.standard_normal(f.shape)
        for fill_open in (False, True):
            f.set_parent(a)
            b = np.asfortranarray(a.copy())
            ocn.fill_halo_regions(f, fill_open_bcs=fill_open, boundary_conditions=_fbcs(ocn, spec))
            g_cpu.fill_halo_regions(b, loc, fill_open, bcs=_oracle_bcs(spec))
            assert np.array_equal(f.parent(), b), (name, fill_open)
    G = m_gpu.tendency("T")
    ga = rng.standard_normal(G.shape)
    G.set_parent(ga)
    gb = np.asfortranarray(ga.copy())
    ocn.compute_flux_bcs(G, _fbcs(ocn, spec_c))
    g_cpu.compute_flux_bcs(gb, (0, 0, 0), _oracle_bcs(spec_c))
    assert np.array_equal(G.parent(), gb)
    with pytest.raises(ValueError):                                     # the array must cover the boundary: (Nx, Ny) points on top
        ocn.fill_halo_regions(flds["T"], boundary_conditions=ocn.FieldBoundaryConditions(top=ocn.ValueBoundaryCondition(np.zeros((3, 3)))))


def test_model_with_array_valued_conditions_matches_oracle(ocn, oracle, arch):
    """the configs[4]-style set-up with spatially varying surface conditions: a wind-stress array on u, a heat-flux array and a
    bottom-gradient array on T, a surface-value array on S; 10 RK3 steps to 1e-12, with the fused substep on (the Flux arrays are
    applied inside the one-pass epilogue) and off (flux_bc_kernel)"""
    size = (16, 12, 10)
    z = tanh_faces(size[2])
    topo = (ocn.Periodic, ocn.Periodic, ocn.Bounded)
    rng = np.random.default_rng(4)
    xy = (size[0], size[1])
    arrs = {("u", "top"): ("Flux", -2e-3 * (1 + 0.5 * rng.standard_normal(xy))), ("T", "top"): ("Flux", 5e-3 * rng.standard_normal(xy)),
            ("T", "bottom"): ("Gradient", 0.4 + 0.1 * rng.standard_normal(xy)), ("S", "top"): ("Value", 35.0 + 0.1 * rng.standard_normal(xy))}
    g_cpu = oracle.Grid(size, topology=(0, 0, 1), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
    m_cpu = oracle.Model(g_cpu, 2)
    cname = {"u": "u", "T": "c0", "S": "c1"}
    for (n, side), (k, a) in arrs.items():
        m_cpu.set_bc(cname[n], side, k.lower(), a)
    results = []
    for fuse in (1, 0):
        g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=topo)
        bcs = {}
        for (n, side), (k, a) in arrs.items():
            bcs.setdefault(n, {})[side] = ocn.BoundaryCondition(k, a)
        m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=("T", "S"), boundary_conditions={n: ocn.FieldBoundaryConditions(**s) for n, s in bcs.items()})
        m_gpu.set_option("fuse_substep", fuse)
        m_gpu.set_option("fused_epilogue", fuse)
        assert m_gpu.get_option("fuse_substep_active") == fuse
        vals = smooth_state({n: g_gpu.nodes(f.loc) for n, f in m_gpu.fields().items()}, 1234)
        ocn.set_model(m_gpu, **vals)
        if fuse:
            m_cpu.set(**{c: vals[n] for c, n in zip(("u", "v", "w", "c0", "c1"), ("u", "v", "w", "T", "S"))})
        dt = 0.1 * g_gpu.Δxᶜᵃᵃ / 0.6
        for _ in range(10):
            ocn.time_step(m_gpu, dt)
            if fuse:
                m_cpu.time_step(dt)
        results.append({n: f.parent() for n, f in m_gpu.fields().items()})
        for name, a, b in field_pairs(m_gpu, m_cpu):
            assert rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]) < 1e-12, (fuse, name)
    for n in results[0]:
        assert np.array_equal(results[0][n], results[1][n]), n


@pytest.mark.parametrize("case", ["ppp_rk3", "ppb_amd_rk3", "ppb_ab2"])
def test_checkpoint_and_restore_continue_bit_identically(ocn, arch, tmp_path, case):
    """write_checkpoint / set_from_checkpoint (`set!(model, filepath)`, OutputWriters/checkpointer.jl:161-231): prognostic fields and
    tendencies with their halos + the clock, in the reference's address layout (container: .npz, see checkpointer.py). A model restored
    in a fresh object continues like the uninterrupted run: bit for bit on the triply periodic grid (RK3 has no hidden state between
    steps); to round-off (1e-13) on Bounded z, where the Fourier-tridiagonal solver -- like the reference's, whose checkpoints do not
    hold it either -- keeps the previous solution in the singular (kx = ky = 0) column that its guarded update re-reads
    (batched_tridiagonal_solver.jl:234-237): the configs[4] physics (diffusivities and hydrostatic pressure are recomputed) and AB2
    (which needs G⁻ and last_Δt from the file)."""
    ppb = case != "ppp_rk3"
    size = (16, 12, 10)
    topo = (ocn.Periodic, ocn.Periodic, ocn.Bounded if ppb else ocn.Periodic)
    z = tanh_faces(size[2]) if ppb else (0.0, 1.0)
    kw = {}
    if case == "ppb_amd_rk3":
        F = ocn.FieldBoundaryConditions
        kw = dict(closure=ocn.AnisotropicMinimumDissipation(), buoyancy=ocn.SeawaterBuoyancy(ocn.LinearEquationOfState(2e-4, 8e-4)),
                  boundary_conditions={"T": F(top=ocn.FluxBoundaryCondition(4e-3)),
                                       "S": F(top=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(b=-2.5e-3), field_dependencies="S"))})
    if case == "ppb_ab2":
        kw = dict(timestepper="QuasiAdamsBashforth2", closure=ocn.ScalarDiffusivity(ν=1e-3, κ=1e-3))

    def make():
        grid = ocn.RectilinearGrid(arch, size=size, x=(0, 1), y=(0, 1), z=z, topology=topo)
        return grid, ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), **kw)

    grid, model = make()
    ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 3))
    dt = 0.05 * grid.Δxᶜᵃᵃ
    for _ in range(3):
        ocn.time_step(model, dt)
    path = ocn.write_checkpoint(model, str(tmp_path / f"checkpoint_iteration{model.clock.iteration}"))
    for _ in range(3):
        ocn.time_step(model, dt)
    want = {n: f.parent() for n, f in model.fields().items()}
    want_clock = (model.clock.time, model.clock.iteration)
    model.close()
    grid2, restored = make()
    ocn.set_from_checkpoint(restored, path)
    assert restored.clock.iteration == 3
    for _ in range(3):
        ocn.time_step(restored, dt)
    assert (restored.clock.time, restored.clock.iteration) == want_clock
    for n, f in restored.fields().items():
        if ppb:
            assert rel_err(f.parent()[3:-3, 3:-3, 3:-3], want[n][3:-3, 3:-3, 3:-3]) < 1e-13, n
        else:
            assert np.array_equal(f.parent(), want[n]), n
    # the file holds the reference's addresses
    with np.load(path) as file:
        keys = set(file.keys())
    assert {"NonhydrostaticModel/u/data", "NonhydrostaticModel/timestepper/Gⁿ/T/data", "NonhydrostaticModel/timestepper/G⁻/w/data",
            "NonhydrostaticModel/clock/time"} <= keys
    other = ocn.RectilinearGrid(arch, size=(8, 8, 8), extent=(1, 1, 1))
    with pytest.raises(ValueError):
        ocn.set_from_checkpoint(ocn.NonhydrostaticModel(grid=other, tracers=("T", "S")), path)


HALO_NS = [(8, 8, 8), (8, 8, 4), (10, 7, 5), (1, 8, 8), (1, 9, 5), (8, 1, 8), (5, 1, 9), (8, 8, 1), (5, 9, 1), (1, 1, 8)]


@pytest.mark.parametrize("N", HALO_NS)
def test_halo_regions_as_the_reference_tests_them(ocn, oracle, arch, N):
    """test/test_halo_regions.jl:1-65 as written: RectilinearGrid(size = N, extent = (100, 200, 300), halo = (1, 1, 1)) -- degenerate
    one-cell directions included --, a CenterField with a random interior: the halos are zero before the fill; after
    fill_halo_regions! on (Periodic, Periodic, Bounded) the x / y halos hold the periodic copies and the z halos the no-flux mirror, `==`.
    (A halo of 1 cannot carry the advection scheme: tendencies and models on such a grid are refused with the reason.)"""
    grid = ocn.RectilinearGrid(arch, size=N, extent=(100, 200, 300), halo=(1, 1, 1), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    assert grid.halo_size == (1, 1, 1)
    field = ocn.CenterField(grid)
    rng = np.random.default_rng(sum(N))
    field.set(rng.random(N))
    d = field.parent()
    Nx, Ny, Nz = N
    assert np.all(d[0] == 0) and np.all(d[-1] == 0) and np.all(d[:, 0] == 0) and np.all(d[:, -1] == 0) and np.all(d[:, :, 0] == 0) and np.all(d[:, :, -1] == 0)
    ocn.fill_halo_regions(field)
    d = field.parent()
    I = (slice(1, 1 + Nx), slice(1, 1 + Ny), slice(1, 1 + Nz))
    assert np.array_equal(d[0:1, I[1], I[2]], d[Nx:Nx + 1, I[1], I[2]])          # data[1-Hx:0] == data[Nx-Hx+1:Nx]
    assert np.array_equal(d[I[0], 0:1, I[2]], d[I[0], Ny:Ny + 1, I[2]])
    assert np.array_equal(d[I[0], I[1], 0:1], d[I[0], I[1], 1:2])                # data[0] == data[1]
    assert np.array_equal(d[I[0], I[1], Nz + 1:Nz + 2], d[I[0], I[1], Nz:Nz + 1])
    # ... and the whole parent array equals the oracle's ordered fills
    g_cpu = oracle.Grid(N, halo=(1, 1, 1), topology=(0, 0, 1), x=(0, 100), y=(0, 200), z=(-300, 0))
    b = g_cpu.zeros((0, 0, 0))
    b[1:-1, 1:-1, 1:-1] = d[1:-1, 1:-1, 1:-1]
    g_cpu.fill_halo_regions(b, (0, 0, 0))
    assert np.array_equal(d, b)
    with pytest.raises(ocn.OcnError, match="halo|size 1"):
        U, G = [ocn.XFaceField(grid), ocn.YFaceField(grid), ocn.ZFaceField(grid)], [ocn.XFaceField(grid), ocn.YFaceField(grid), ocn.ZFaceField(grid)]
        ocn.kernels.compute_tendencies(grid, U[0], U[1], U[2], [], G[0], G[1], G[2], [])


@pytest.mark.parametrize("topology", TOPOS)
@pytest.mark.parametrize("ntracers", [0, 1, 3, 4, 8])
def test_tracer_counts_match_oracle(ocn, oracle, arch, topology, ntracers):
    """the model with 0, 1, 3 (the one-field-per-workgroup kernel's instantiations), 4 and 8 tracers (more than that kernel takes: the
    per-field kernels; 8 = the library's maximum): tendencies bit-identical to the oracle for every implementation that accepts the
    count, fields within 1e-12 after 3 RK3 steps"""
    size = (16, 12, 10)
    z = tanh_faces(size[2]) if topology[2] == "Bounded" else None
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=z, ntracers=ntracers)
    set_both(ocn, m_gpu, m_cpu, seed=5, enforce_incompressibility=False)
    m_cpu.update_state(True)
    cpu_names = ["u", "v", "w"] + ["c%d" % t for t in range(ntracers)]
    for impl in (0, 1, 2):
        m_gpu.set_option("tendency_impl", impl)
        for n in m_gpu.fields():
            m_gpu.tendency(n).set_parent(np.zeros(m_gpu.tendency(n).shape))
        ocn.update_state(m_gpu, True)
        for n, cn in zip(m_gpu.fields().keys(), cpu_names):
            assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), (impl, n)
    m_gpu.set_option("tendency_impl", 2)
    set_both(ocn, m_gpu, m_cpu, seed=6, smooth=True)
    dt = 0.1 * g_gpu.Δxᶜᵃᵃ / 0.6
    for _ in range(3):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    for name, a, b in field_pairs(m_gpu, m_cpu):
        assert rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]) < 1e-12, (name, ntracers)


@pytest.mark.parametrize("size", [(65, 8, 6), (129, 15, 5), (64, 7, 7), (63, 14, 4), (130, 6, 9), (200, 3, 3)])
@pytest.mark.parametrize("topology", TOPOS)
def test_tile_edges_of_the_tendency_kernels(ocn, oracle, arch, size, topology):
    """sizes around the 64 x 7 tiles of the one-field-per-workgroup kernel (one column / one row more or fewer than whole tiles, a single
    row of tiles, Ny = 3): every implementation bit-identical to the oracle, also with the fused substep (3 steps within 1e-12)"""
    z = tanh_faces(size[2]) if topology[2] == "Bounded" else None
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=z)
    set_both(ocn, m_gpu, m_cpu, seed=9, enforce_incompressibility=False)
    m_cpu.update_state(True)
    for impl in (0, 1, 2):
        m_gpu.set_option("tendency_impl", impl)
        for n in m_gpu.fields():
            m_gpu.tendency(n).set_parent(np.zeros(m_gpu.tendency(n).shape))
        ocn.update_state(m_gpu, True)
        for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
            assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), (impl, n)
    m_gpu.set_option("tendency_impl", 2)
    set_both(ocn, m_gpu, m_cpu, seed=10, smooth=True)
    dt = 0.1 * min(g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ) / 0.6
    for _ in range(3):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    umax = max(np.abs(m_cpu.field(n)).max() for n in ("u", "v", "w"))
    dmax = max(g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, float(np.max(g_gpu.Δzᵃᵃᶜ)))
    for name, a, b in field_pairs(m_gpu, m_cpu):
        ia, ib = a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]
        if name == "pNHS":
            # strongly anisotropic cells (129 x 15 x 5 on the unit cube): p = lap^-1(div u*) / dt carries the velocities' round-off times
            # dx / dt -- the error model of test_adapted_advection_order_matches_oracle
            pscale = max(np.abs(ib).max(), umax * dmax / dt)
            assert np.max(np.abs(ia - ib)) < 1e-12 * pscale, (name, size, np.max(np.abs(ia - ib)), pscale)
            continue
        assert rel_err(ia, ib) < 1e-12, (name, size)


@pytest.mark.parametrize("size,topology", [((64, 64, 64), TOPOS[0]), ((96, 80, 40), TOPOS[1]), ((128, 49, 33), TOPOS[0])])
def test_mid_size_parity_with_the_oracle(ocn, oracle, arch, size, topology):
    """the largest sizes the oracle finishes in seconds: several tiles and workgroups per direction, the z march over chunks, the fused
    substep, the split pressure solve (64^3) and the library-plan solve (other sizes) -- tendencies bit-identical, 5 RK3 steps within 1e-12"""
    z = tanh_faces(size[2]) if topology[2] == "Bounded" else None
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=z)
    set_both(ocn, m_gpu, m_cpu, seed=21, enforce_incompressibility=False)
    m_cpu.update_state(True)
    for impl in (1, 2):
        m_gpu.set_option("tendency_impl", impl)
        ocn.update_state(m_gpu, True)
        for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
            assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), (impl, n)
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.1 * min(g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ) / 0.6
    for _ in range(5):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    assert m_gpu.clock.time == m_cpu.time
    umax = max(np.abs(m_cpu.field(n)).max() for n in ("u", "v", "w"))
    dmax = max(g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, float(np.max(g_gpu.Δzᵃᵃᶜ)))
    for name, a, b in field_pairs(m_gpu, m_cpu):
        ia, ib = a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]
        if name == "pNHS":       # stretched cells: the error model of test_adapted_advection_order_matches_oracle (eps |u| dx / dt reaches p)
            assert np.max(np.abs(ia - ib)) < 1e-12 * max(np.abs(ib).max(), umax * dmax / dt), (name, size)
            continue
        assert rel_err(ia, ib) < 1e-12, (name, size, rel_err(ia, ib))
    assert ocn.max_abs_divergence(m_gpu) < 5e-8


def test_boundary_conditions_on_the_diffusivity_fields_match_oracle(ocn, oracle, arch):
    """boundary_conditions = (νₑ = ..., κₑ = (T = ...,)) of an AnisotropicMinimumDissipation model (anisotropic_minimum_dissipation.jl:
    339-352: the diffusivity fields are CenterFields built with the user's conditions, filled after compute_diffusivities!): Value and
    Gradient conditions at bottom / top; diffusivity fields with their halos and the tendencies bit-identical to the oracle, 5 steps 1e-12.
    The reference's own test of this feature runs in tests/test_gpu_reference_tests.py (fluxes_with_diffusivity_boundary_conditions)."""
    size = (12, 10, 8)
    z = tanh_faces(size[2])
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    g_cpu = oracle.Grid(size, topology=(0, 0, 1), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
    F = ocn.FieldBoundaryConditions
    bcs = {"νₑ": F(bottom=ocn.ValueBoundaryCondition(2e-3)), "κₑ": {"T": F(bottom=ocn.ValueBoundaryCondition(5e-3), top=ocn.GradientBoundaryCondition(-1e-2))},
           "T": F(bottom=ocn.GradientBoundaryCondition(0.4))}
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=("T", "S"), closure=ocn.AnisotropicMinimumDissipation(), boundary_conditions=bcs)
    m_cpu = oracle.Model(g_cpu, 2)
    m_cpu.set_amd(C=1 / 3, Ckappa=[1 / 3, 1 / 3])
    m_cpu.set_bc("nu_e", "bottom", "value", 2e-3)
    m_cpu.set_bc("kappa_e0", "bottom", "value", 5e-3)
    m_cpu.set_bc("kappa_e0", "top", "gradient", -1e-2)
    m_cpu.set_bc("c0", "bottom", "gradient", 0.4)
    set_both(ocn, m_gpu, m_cpu, seed=8, enforce_incompressibility=False)
    for fused in (1, 0):
        m_gpu.set_option("fused_epilogue", fused)
        ocn.update_state(m_gpu, True)
        m_cpu.update_state(True)
        D = m_gpu.diffusivity_fields
        assert np.array_equal(D[0].parent(), m_cpu.field("nu_e")) and np.array_equal(D[1][0].parent(), m_cpu.field("kappa_e0"))
        assert np.array_equal(D[1][1].parent(), m_cpu.field("kappa_e1"))
        nu = D[0].parent()
        assert np.allclose(0.5 * (nu[3:-3, 3:-3, 2] + nu[3:-3, 3:-3, 3]), 2e-3, rtol=1e-14)        # the Value condition sits on the bottom face
        for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
            assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), (n, fused)
    m_gpu.set_option("fused_epilogue", 1)
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.05 * min(g_gpu.Δxᶜᵃᵃ, float(np.min(g_gpu.Δzᵃᵃᶜ))) / 0.6
    for _ in range(5):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    for name, a, b in field_pairs(m_gpu, m_cpu):
        assert rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]) < 1e-12, name
    with pytest.raises(ocn.OcnError):
        ocn.NonhydrostaticModel(grid=g_gpu, closure=ocn.AnisotropicMinimumDissipation(), boundary_conditions={"νₑ": F(bottom=ocn.FluxBoundaryCondition(1.0))})


def test_seeded_random_physics_configurations_match_the_oracle(ocn, oracle, arch):
    """sixteen seeded random small models against the ORACLE: random sizes (4 .. 20 per direction), Periodic / Bounded mixes (stretched z on
    half of the Bounded ones), 0 .. 2 tracers, ScalarDiffusivity or AnisotropicMinimumDissipation, FPlane, buoyancy (tracer or linear
    seawater), valued Flux conditions on random walls and a field-dependent one. Tendencies from identical inputs (no projection) bit for
    bit -- every physics kernel, the z-marching ones included --, then 3 RK3 steps from a projected smooth state within 1e-12."""
    rng = np.random.default_rng(771)
    F = ocn.FieldBoundaryConditions
    topo_names = ("Periodic", "Bounded")
    sides = {0: ("west", "east"), 1: ("south", "north"), 2: ("bottom", "top")}
    normal = {"u": 0, "v": 1, "w": 2}
    for case in range(16):
        size = tuple(int(rng.integers(4, 21)) for _ in range(3))
        topology = tuple(topo_names[int(rng.random() < 0.5)] for _ in range(3))
        z = tanh_faces(size[2]) if (topology[2] == "Bounded" and rng.random() < 0.5) else ((-1.0, 0.0) if topology[2] == "Bounded" else (0.0, 1.0))
        ntr = int(rng.integers(0, 3))
        gpu_names = ("T", "S")[:ntr]
        cpu_names = ["u", "v", "w"] + ["c%d" % t for t in range(ntr)]
        amd = rng.random() < 0.5
        kw = dict(closure=ocn.AnisotropicMinimumDissipation() if amd else ocn.ScalarDiffusivity(ν=3e-3, κ=2e-3))
        g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=tuple(getattr(ocn, t) for t in topology))
        g_cpu = oracle.Grid(size, topology=tuple({"Periodic": 0, "Bounded": 1}[t] for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
        m_cpu = oracle.Model(g_cpu, ntr)
        if amd:
            m_cpu.set_amd()
        else:
            m_cpu.set_closure(nu=3e-3, kappa=2e-3)
        if rng.random() < 0.5:
            kw["coriolis"] = ocn.FPlane(f=0.6)
            m_cpu.set_coriolis(0.6)
        if ntr == 2 and rng.random() < 0.6:
            kw["buoyancy"] = ocn.SeawaterBuoyancy(ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4))
            m_cpu.set_seawater_buoyancy(alpha=2e-4, beta=8e-4)
        elif ntr >= 1 and rng.random() < 0.4:
            gpu_names = ("b",) + gpu_names[1:]
            kw["buoyancy"] = ocn.BuoyancyTracer()
            m_cpu.set_buoyancy_tracer(0)
        bcs = {}
        for name, cname in zip(("u", "v", "w") + gpu_names, cpu_names):
            conds = {}
            for d in range(3):
                if topology[d] == "Bounded" and normal.get(name) != d:
                    for sd in sides[d]:
                        if rng.random() < 0.3:
                            val = float(rng.normal()) * 1e-3
                            conds[sd] = ocn.FluxBoundaryCondition(val)
                            m_cpu.set_bc(cname, sd, "flux", val)
            if conds:
                bcs[name] = F(**conds)
        if ntr >= 1 and topology[2] == "Bounded" and gpu_names[-1] not in bcs and rng.random() < 0.5:
            rate = 2.5e-3
            bcs[gpu_names[-1]] = F(top=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(b=-rate), field_dependencies=gpu_names[-1]))
            m_cpu.set_linear_flux_bc(cpu_names[-1], "top", 0.0, -rate, cpu_names[-1])
        if bcs:
            kw["boundary_conditions"] = bcs
        m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=gpu_names, **kw)
        what = (case, size, topology, ntr, sorted(kw), sorted(bcs))
        set_both(ocn, m_gpu, m_cpu, seed=900 + case, enforce_incompressibility=False)
        ocn.update_state(m_gpu, True)
        m_cpu.update_state(True)
        for n, cn in zip(m_gpu.fields().keys(), cpu_names):
            assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), (what, n)
        set_both(ocn, m_gpu, m_cpu, seed=1234 + case, smooth=True)
        dz = float(np.min(g_gpu.Δzᵃᵃᶜ))
        dt = 0.05 * min(g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, dz) / 0.6
        for _ in range(3):
            ocn.time_step(m_gpu, dt)
            m_cpu.time_step(dt)
        for name, a, b in field_pairs(m_gpu, m_cpu):
            if name == "pNHS":
                continue                                    # the pressure has its own error model (test_time_step_parity_10_steps)
            e = rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3])
            assert e < 1e-12, (what, name, e)
        m_gpu.close()


def test_flux_condition_changed_between_steps_acts_on_the_next_first_stage(ocn, oracle, arch):
    """compute_flux_bc_tendencies! belongs to the STAGE (runge_kutta_3.jl:118,134,150: right before rk3_substep!), not to update_state!: after a
    time-step the stored tendencies carry no Flux-condition terms, and a condition whose value changes between two steps (a wind stress
    updated by a callback) acts on the next step's first stage with its NEW value. The library folds the conditions into the tendency pass
    only where the next stage's substep rides along; here: tendencies after update_state! and after a step bit / 1e-10 against the oracle,
    and a run that switches the surface stress after step 2 against the oracle doing the same, with the fused and the separate substeps"""
    from oldoceananigans_jl_amd import _lib
    size, topology = (12, 10, 9), ("Periodic", "Periodic", "Bounded")
    z = tanh_faces(size[2])
    F = ocn.FieldBoundaryConditions
    for fuse in (1, 0):
        g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=tuple(getattr(ocn, t) for t in topology))
        g_cpu = oracle.Grid(size, topology=(0, 0, 1), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
        m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=("T", "S"), closure=ocn.ScalarDiffusivity(ν=2e-3, κ=1e-3),
                                        boundary_conditions={"u": F(top=ocn.FluxBoundaryCondition(-1e-3)), "T": F(top=ocn.FluxBoundaryCondition(4e-3))})
        m_gpu.set_option("fuse_substep", fuse)
        m_cpu = oracle.Model(g_cpu, 2)
        m_cpu.set_closure(nu=2e-3, kappa=1e-3)
        m_cpu.set_bc("u", "top", "flux", -1e-3)
        m_cpu.set_bc("c0", "top", "flux", 4e-3)
        set_both(ocn, m_gpu, m_cpu, seed=3, enforce_incompressibility=False)
        ocn.update_state(m_gpu, True)
        m_cpu.update_state(True)
        for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
            assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), (fuse, n)     # no Flux terms in update_state!'s tendencies
        set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
        dt = 2e-3
        for step in range(4):
            if step == 2:                                    # the stress reverses between two steps
                _lib.check(_lib.lib().ocn_model_set_boundary_condition(m_gpu.handle, b"u", 5, 1, 2.5e-3))
                m_cpu.set_bc("u", "top", "flux", 2.5e-3)
            ocn.time_step(m_gpu, dt)
            m_cpu.time_step(dt)
            for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
                a, b = m_gpu.tendency(n).parent()[3:-3, 3:-3, 3:-3], m_cpu.field("G" + cn)[3:-3, 3:-3, 3:-3]
                assert rel_err(a, b) < 1e-10, (fuse, step, n, rel_err(a, b))
        for name, a, b in field_pairs(m_gpu, m_cpu):
            if name != "pNHS":
                assert rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]) < 1e-12, (fuse, name)
        m_gpu.close()


def test_seeded_random_flat_ab2_and_value_gradient_configurations_match_the_oracle(ocn, oracle, arch):
    """sixteen more seeded random small models against the ORACLE, on the axes the first random test leaves out: Flat directions (at most one,
    size 1), Value / Gradient conditions on walls (no-slip velocities, fixed tracers), the QuasiAdamsBashforth2 stepper on half of the
    cases, with ScalarDiffusivity / FPlane / buoyancy mixes. Tendencies from identical inputs bit for bit, then 3 steps within 1e-12."""
    rng = np.random.default_rng(4242)
    F = ocn.FieldBoundaryConditions
    sides = {0: ("west", "east"), 1: ("south", "north"), 2: ("bottom", "top")}
    normal = {"u": 0, "v": 1, "w": 2}
    code = {"Periodic": 0, "Bounded": 1, "Flat": 3}
    for case in range(16):
        topology = [("Periodic", "Bounded")[int(rng.random() < 0.5)] for _ in range(3)]
        flat = int(rng.integers(0, 4))                      # 3: no Flat direction
        if flat < 3:
            topology[flat] = "Flat"
        topology = tuple(topology)
        size = tuple(1 if t == "Flat" else int(rng.integers(5, 19)) for t in topology)
        z = tanh_faces(size[2]) if (topology[2] == "Bounded" and rng.random() < 0.5) else ((-1.0, 0.0) if topology[2] == "Bounded" else (0.0, 1.0))
        ntr = int(rng.integers(1, 3))
        gpu_names = ("T", "S")[:ntr]
        cpu_names = ["u", "v", "w"] + ["c%d" % t for t in range(ntr)]
        ab2 = rng.random() < 0.5
        kw = dict(timestepper="QuasiAdamsBashforth2" if ab2 else "RungeKutta3")
        g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=tuple(getattr(ocn, t) for t in topology))
        g_cpu = oracle.Grid(size, topology=tuple(code[t] for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
        m_cpu = oracle.Model(g_cpu, ntr)
        if rng.random() < 0.7:
            kw["closure"] = ocn.ScalarDiffusivity(ν=3e-3, κ=2e-3)
            m_cpu.set_closure(nu=3e-3, kappa=2e-3)
        if rng.random() < 0.5:
            kw["coriolis"] = ocn.FPlane(f=0.6)
            m_cpu.set_coriolis(0.6)
        if ntr == 2 and topology[2] != "Flat" and rng.random() < 0.5:
            kw["buoyancy"] = ocn.SeawaterBuoyancy()
            m_cpu.set_seawater_buoyancy()
        bcs = {}
        for name, cname in zip(("u", "v", "w") + gpu_names, cpu_names):
            conds = {}
            for d in range(3):
                if topology[d] == "Bounded" and normal.get(name) != d:
                    for sd in sides[d]:
                        r = rng.random()
                        val = float(rng.normal()) * (1e-2 if name in ("u", "v", "w") else 0.3)
                        if r < 0.25:
                            conds[sd] = ocn.ValueBoundaryCondition(val); m_cpu.set_bc(cname, sd, "value", val)
                        elif r < 0.45:
                            conds[sd] = ocn.GradientBoundaryCondition(val); m_cpu.set_bc(cname, sd, "gradient", val)
                        elif r < 0.6:
                            conds[sd] = ocn.FluxBoundaryCondition(val * 1e-1); m_cpu.set_bc(cname, sd, "flux", val * 1e-1)
            if conds:
                bcs[name] = F(**conds)
        if bcs:
            kw["boundary_conditions"] = bcs
        m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=gpu_names, **kw)
        what = (case, size, topology, ntr, sorted(kw), {n: sorted(b.sides) for n, b in bcs.items()})
        set_both(ocn, m_gpu, m_cpu, seed=500 + case, enforce_incompressibility=False)
        ocn.update_state(m_gpu, True)
        m_cpu.update_state(True)
        for n, cn in zip(m_gpu.fields().keys(), cpu_names):
            assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), (what, n)
        set_both(ocn, m_gpu, m_cpu, seed=77 + case, smooth=True)
        spacings = [d for d, t in zip((g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, float(np.min(g_gpu.Δzᵃᵃᶜ))), topology) if t != "Flat"]
        dt = 0.05 * min(spacings) / 0.6
        for _ in range(3):
            ocn.time_step(m_gpu, dt)
            if ab2:
                m_cpu.time_step_ab2(dt)
            else:
                m_cpu.time_step(dt)
        core = tuple(slice(None) if t == "Flat" else slice(3, -3) for t in topology)
        for name, a, b in field_pairs(m_gpu, m_cpu):
            if name == "pNHS":
                continue
            e = rel_err(a[core], b[core])
            assert e < 1e-12, (what, name, e)
        m_gpu.close()


def test_api_call_sequences_between_steps_match_the_oracle(ocn, oracle, arch):
    """what happens BETWEEN time-steps follows the reference too: set! in the middle of a run ends with update_state!(compute_tendencies = false)
    (set_nonhydrostatic_model.jl:52-57), so the next first stage still uses the tendencies of the state before the set! (only iteration 0
    re-evaluates them, runge_kutta_3.jl:97); an explicit update_state! re-evaluates them. Sequence: 2 steps, set! of a new state, 2 steps,
    update_state!, 1 step -- HIP against the oracle making the same calls, fields within 1e-12 after every call."""
    size, topology = (12, 10, 8), ("Periodic", "Bounded", "Bounded")
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=tanh_faces(size[2]))

    def check(tag):
        for name, a, b in field_pairs(m_gpu, m_cpu):
            if name != "pNHS":
                assert rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]) < 1e-12, (tag, name, rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3]))

    set_both(ocn, m_gpu, m_cpu, seed=11, smooth=True)
    dt = 2e-3
    for _ in range(2):
        ocn.time_step(m_gpu, dt); m_cpu.time_step(dt)
    check("two steps")
    set_both(ocn, m_gpu, m_cpu, seed=12, smooth=True)             # a new state in the middle of the run
    check("set!")
    for _ in range(2):
        ocn.time_step(m_gpu, dt); m_cpu.time_step(dt)
    check("two steps after set!")
    ocn.update_state(m_gpu, True); m_cpu.update_state(True)
    ocn.time_step(m_gpu, dt); m_cpu.time_step(dt)
    check("step after update_state!")
    assert m_gpu.clock.iteration == 5 and abs(m_gpu.clock.time - m_cpu.time) < 1e-15
    m_gpu.close()


def test_ab2_with_a_time_step_that_changes_matches_the_oracle(ocn, oracle, arch):
    """QuasiAdamsBashforth2: the first step and every step whose Δt differs from clock.last_Δt are forward-Euler steps (χ = -1/2,
    quasi_adams_bashforth_2.jl:86-97); a Flux condition and a closure ride along. Δt sequence dt, dt, 0.7 dt, 0.7 dt, dt against the oracle."""
    size, topology = (10, 12, 9), ("Bounded", "Periodic", "Bounded")
    z = tanh_faces(size[2])
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=tuple(getattr(ocn, t) for t in topology))
    g_cpu = oracle.Grid(size, topology=(1, 0, 1), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
    F = ocn.FieldBoundaryConditions
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=("T", "S"), timestepper="QuasiAdamsBashforth2", closure=ocn.ScalarDiffusivity(ν=2e-3, κ=1e-3),
                                    boundary_conditions={"T": F(top=ocn.FluxBoundaryCondition(3e-3)), "v": F(west=ocn.ValueBoundaryCondition(0.0))})
    m_cpu = oracle.Model(g_cpu, 2)
    m_cpu.set_closure(nu=2e-3, kappa=1e-3)
    m_cpu.set_bc("c0", "top", "flux", 3e-3)
    m_cpu.set_bc("v", "west", "value", 0.0)
    set_both(ocn, m_gpu, m_cpu, seed=21, smooth=True)
    dt = 1.5e-3
    for n, step_dt in enumerate([dt, dt, 0.7 * dt, 0.7 * dt, dt]):
        ocn.time_step(m_gpu, step_dt)
        m_cpu.time_step_ab2(step_dt)
        for name, a, b in field_pairs(m_gpu, m_cpu):
            if name != "pNHS":
                e = rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3])
                assert e < 1e-12, (n, name, e)
    assert abs(m_gpu.clock.time - m_cpu.time) < 1e-15
    m_gpu.close()


def test_seeded_random_launch_ranges_of_the_flux_sharing_tendency_kernels(ocn, arch):
    """KernelParameters launches (kernel_launching.jl:25-95: the interior / strip ranges of the distributed update, a29-a30): twenty seeded
    random index ranges on a triply periodic and on a stretched Bounded-z grid, the flux-sharing kernels (role kernel, all-fields kernel:
    ocn_compute_tendencies with tendency_impl 2 and 1) against the per-field kernels (ocn_compute_Gu .. Gc, bit-identical to the oracle)
    on the same range: equal inside the range, nothing written outside it (the arrays carry a sentinel)"""
    rng = np.random.default_rng(99)
    for topology, size in ((("Periodic", "Periodic", "Periodic"), (70, 12, 20)), (("Periodic", "Periodic", "Bounded"), (66, 9, 14))):
        z = tanh_faces(size[2]) if topology[2] == "Bounded" else (0.0, 1.0)
        grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=tuple(getattr(ocn, t) for t in topology))
        u, v, w, T, S = ocn.XFaceField(grid), ocn.YFaceField(grid), ocn.ZFaceField(grid), ocn.CenterField(grid), ocn.CenterField(grid)
        vals = smooth_state({n: grid.nodes(f.loc) for n, f in zip("uvwTS", (u, v, w, T, S))}, 5)
        for n, f in zip("uvwTS", (u, v, w, T, S)):
            f.set(vals[n])
            ocn.fill_halo_regions(f)
        mk = {"u": ocn.XFaceField, "v": ocn.YFaceField, "w": ocn.ZFaceField, "T": ocn.CenterField, "S": ocn.CenterField}
        for case in range(10):
            lo = [int(rng.integers(1, n + 1)) for n in size]
            hi = [int(rng.integers(l, n + 1)) for l, n in zip(lo, size)]
            if case == 0:
                lo, hi = [1, 1, 1], list(size)
            r = (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])
            want = {n: mk[n](grid) for n in "uvwTS"}
            for f in want.values():
                f.set_parent(np.full(f.shape, -7.25))
            ocn.kernels.compute_Gu(grid, u, v, w, want["u"], kernel_parameters=r)
            ocn.kernels.compute_Gv(grid, u, v, w, want["v"], kernel_parameters=r)
            ocn.kernels.compute_Gw(grid, u, v, w, want["w"], kernel_parameters=r)
            ocn.kernels.compute_Gc(grid, u, v, w, T, want["T"], kernel_parameters=r)
            ocn.kernels.compute_Gc(grid, u, v, w, S, want["S"], kernel_parameters=r)
            for impl in (2, 1):
                ocn.set_option("tendency_impl", impl)
                try:
                    got = {n: mk[n](grid) for n in "uvwTS"}
                    for f in got.values():
                        f.set_parent(np.full(f.shape, -7.25))
                    ocn.kernels.compute_tendencies(grid, u, v, w, [T, S], got["u"], got["v"], got["w"], [got["T"], got["S"]], kernel_parameters=r)
                finally:
                    ocn.set_option("tendency_impl", 2)
                for n in "uvwTS":
                    a, b = got[n].parent(), want[n].parent()
                    assert np.array_equal(a, b), (topology, r, impl, n, int((a != b).sum()))
                    assert (b != -7.25).sum() > 0


@pytest.mark.parametrize("topology", [("Periodic", "Periodic", "Periodic"), ("Periodic", "Periodic", "Bounded")])
def test_stage_pressures_that_nothing_can_read_are_not_stored(ocn, arch, topology):
    """RK3 stages 1 and 2: the stage's pNHS is overwritten by the next stage before anything outside the time-step call can read it, so the
    dense-solution path does not store it (nor fill its halos) -- option skip_stage_pressure = 0 stores it like the reference does. Fields,
    tendencies and the pressure left after each of three steps (the last stage's, halos included) bit for bit."""
    size = (32, 16, 16)
    z = tanh_faces(size[2]) if topology[2] == "Bounded" else (0.0, 1.0)
    grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=tuple(getattr(ocn, t) for t in topology))
    outs = []
    for skip in (1, 0):
        ocn.set_option("skip_stage_pressure", skip)
        try:
            model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
            ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 8))
            per_step = []
            for _ in range(3):
                ocn.time_step(model, 1e-3)
                per_step.append([f.parent() for f in model.fields().values()] + [model.tendency(n).parent() for n in model.fields()] + [model.pressures.pNHS.parent()])
            outs.append(per_step)
            model.close()
        finally:
            ocn.set_option("skip_stage_pressure", 1)
    for a, b in zip(outs[0], outs[1]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert np.abs(outs[0][-1][-1]).max() > 0


@pytest.mark.parametrize("case", ["ppp", "ppb_amd_flux", "bbb_scalar"])
def test_the_tendency_after_the_second_stage_is_not_stored_and_nobody_can_tell(ocn, arch, case):
    """G(U²), evaluated after RK3's second stage, feeds the third stage's substep (riding along in the same kernel) and nothing else: no
    cache_previous_tendencies! follows the third stage and the closing update_state! overwrites Gⁿ (runge_kutta_3.jl:150-166). The kernels
    that carry that substep therefore do not store it (option skip_dead_tendency_store = 0 stores it). After each of three steps: fields, Gⁿ
    (= G(U³)), G⁻ (= G(U¹)) and the pressure bit for bit."""
    F = ocn.FieldBoundaryConditions
    kw = {}
    if case == "ppp":
        grid = ocn.RectilinearGrid(arch, size=(70, 12, 16), extent=(1, 1, 1))
    elif case == "ppb_amd_flux":
        grid = ocn.RectilinearGrid(arch, size=(40, 12, 14), x=(0, 1), y=(0, 1), z=tanh_faces(14), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
        kw = dict(closure=ocn.AnisotropicMinimumDissipation(), buoyancy=ocn.SeawaterBuoyancy(), coriolis=ocn.FPlane(f=0.3),
                  boundary_conditions={"u": F(top=ocn.FluxBoundaryCondition(-1e-3)),
                                       "S": F(top=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(b=-2.5e-3), field_dependencies="S"))})
    else:
        grid = ocn.RectilinearGrid(arch, size=(20, 11, 9), extent=(1, 1, 1), topology=(ocn.Bounded,) * 3)
        kw = dict(closure=ocn.ScalarDiffusivity(ν=2e-3, κ=1e-3))
    outs = []
    for skip in (1, 0):
        ocn.set_option("skip_dead_tendency_store", skip)
        try:
            model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), **kw)
            ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 31))
            per_step = []
            for _ in range(3):
                ocn.time_step(model, 1e-3)
                per_step.append([f.parent() for f in model.fields().values()] + [model.tendency(n).parent() for n in model.fields()] +
                                [model.tendency(n, previous=True).parent() for n in model.fields()] + [model.pressures.pNHS.parent()])
            outs.append(per_step)
            model.close()
        finally:
            ocn.set_option("skip_dead_tendency_store", 1)
    for a, b in zip(outs[0], outs[1]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
