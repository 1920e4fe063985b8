"""GPU: several solvers / models of different sizes in ONE process. rocFFT can return wrong transforms from a new multi-dimensional
plan while plans of other sizes are alive (tools/fft_real_test2.hip: e.g. the 64x16x8 real 3-D pair after 32x16x8, 8x16x32, 32^3).
Triage (same tool): the unit-stride batched 1-D complex plans stay exact in that situation. The library (i) releases plans as soon
as a model dies (no reference cycles in the host mirror), (ii) verifies every plan set at creation and (iii) switches a solver whose
multi-dimensional plans fail that check to the per-direction path on 1-D plans -- so every creation succeeds and every solve is
correct: no refusal, never a silently wrong pressure."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SIZES = ((32, 16, 8), (16, 16, 16), (8, 16, 32), (32, 32, 32), (64, 16, 8), (32, 16, 16))


def _real_vs_c2c(ocn, arch, size, rng):
    grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=(0.0, 1.0))
    u, v, w = ocn.XFaceField(grid), ocn.YFaceField(grid), ocn.ZFaceField(grid)
    for f in (u, v, w):
        f.set(rng.standard_normal(size))
    ocn.fill_halo_regions([u, v, w])
    solver = ocn.FFTBasedPoissonSolver(grid)
    p_real, p_c2c = ocn.CenterField(grid), ocn.CenterField(grid)
    try:
        ocn.set_option("real_fft", 1)
        ocn.solve_for_pressure(p_real, solver, (u, v, w))
        ocn.set_option("real_fft", 0)
        ocn.solve_for_pressure(p_c2c, solver, (u, v, w))
    finally:
        ocn.set_option("real_fft", 1)
    a, b = p_real.parent()[3:-3, 3:-3, 3:-3], p_c2c.parent()[3:-3, 3:-3, 3:-3]
    return np.abs(a - b).max() / np.abs(b).max(), solver


def test_sequential_solvers_of_different_sizes_are_all_correct(ocn, arch):
    rng = np.random.default_rng(0)
    for size in SIZES + SIZES[::-1]:
        err, solver = _real_vs_c2c(ocn, arch, size, rng)
        assert err < 1e-12, (size, err)
        del solver


def test_live_solvers_of_different_sizes_are_all_created_and_all_correct(ocn, arch):
    from oldoceananigans_jl_amd import _lib
    rng = np.random.default_rng(1)
    before = _lib.lib().ocn_debug_fft_fallbacks()
    keep = []
    for size in SIZES + ((24, 20, 12), (48, 16, 8), (64, 32, 8)):
        err, solver = _real_vs_c2c(ocn, arch, size, rng)          # raises OcnError if a solver is refused: none may be
        keep.append(solver)
        assert err < 1e-12, (size, err)
    # solvers created earlier still solve correctly while the later ones are alive
    for size, solver in zip(SIZES, keep):
        grid = solver.grid
        u, v, w = ocn.XFaceField(grid), ocn.YFaceField(grid), ocn.ZFaceField(grid)
        for f in (u, v, w):
            f.set(rng.standard_normal(size))
        ocn.fill_halo_regions([u, v, w])
        p = ocn.CenterField(grid)
        ocn.solve_for_pressure(p, solver, (u, v, w))
        ocn.fill_halo_regions(p)
        a = p.parent()
        lap = ((a[4:-2, 3:-3, 3:-3] - 2 * a[3:-3, 3:-3, 3:-3] + a[2:-4, 3:-3, 3:-3]) / grid.Δxᶜᵃᵃ ** 2 +
               (a[3:-3, 4:-2, 3:-3] - 2 * a[3:-3, 3:-3, 3:-3] + a[3:-3, 2:-4, 3:-3]) / grid.Δyᵃᶜᵃ ** 2 +
               (a[3:-3, 3:-3, 4:-2] - 2 * a[3:-3, 3:-3, 3:-3] + a[3:-3, 3:-3, 2:-4]) / grid.Δzᵃᵃᶜ[0] ** 2)
        U, V, W = u.parent(), v.parent(), w.parent()
        div = ((U[4:-2, 3:-3, 3:-3] - U[3:-3, 3:-3, 3:-3]) / grid.Δxᶜᵃᵃ + (V[3:-3, 4:-2, 3:-3] - V[3:-3, 3:-3, 3:-3]) / grid.Δyᵃᶜᵃ +
               (W[3:-3, 3:-3, 4:-2] - W[3:-3, 3:-3, 3:-3]) / grid.Δzᵃᵃᶜ[0])
        assert np.abs(lap - div).max() < 1e-9 * np.abs(div).max(), size
    print("solvers on the per-direction fallback:", _lib.lib().ocn_debug_fft_fallbacks() - before)
    for s in keep:
        s.close()
