"""GPU: several solvers / models of different sizes in ONE process. rocFFT can return wrong transforms from a new plan while
plans of other sizes are alive (tools/fft_real_test2.hip); the library (i) releases plans as soon as a model dies (no reference
cycles in the host mirror) and (ii) verifies every plan set at creation, so the outcome is either a correct solve or a LOUD
OcnError (status OCN_EFFT) -- never a silently wrong pressure."""
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


def test_live_solvers_of_different_sizes_never_give_silently_wrong_results(ocn, arch):
    rng = np.random.default_rng(1)
    keep, refused = [], 0
    for size in SIZES:
        try:
            err, solver = _real_vs_c2c(ocn, arch, size, rng)
        except ocn.OcnError as e:
            assert "self-check" in str(e), e
            refused += 1
            continue
        keep.append(solver)
        assert err < 1e-12, (size, err)
    for s in keep:
        s.close()
    assert refused < len(SIZES)
