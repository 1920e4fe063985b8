"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances: halo / index work is bit-exact (`==`, like test/test_halo_regions.jl:22-41). Floating-point fields: 1e-12
relative (BASELINE.json north_star) after 10 RK3 steps; single kernels are expected to be bit-identical to the oracle
because both sides evaluate the same IEEE operation sequence (FMA only at the reference's @muladd sites)."""
import numpy as np
import pytest

from helpers import field_pairs, make_pair, rel_err, set_both, tanh_faces

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
    for impl in (0, 1):
        m_gpu.set_option("tendency_impl", impl)
        ocn.update_state(m_gpu, True)
        for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
            G_gpu = m_gpu.tendency(n).parent()
            G_cpu = m_cpu.field("G" + cn)
            assert np.array_equal(G_gpu, G_cpu), (impl, n, np.abs(G_gpu - G_cpu).max())


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
        assert rel_err(got, p_cpu[3:-3, 3:-3, 3:-3]) < 1e-11, (size, rel_err(got, p_cpu[3:-3, 3:-3, 3:-3]))
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
