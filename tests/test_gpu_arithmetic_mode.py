"""GPU: the opt-in CONTRACTED arithmetic of the WENO-5 flux (option "arithmetic" = 1, ocn_device.h) -- what north_star's 1e-12 tolerance
buys on the headline kernel, measured rather than assumed. The default (0) stays the reference's IEEE operation sequence and is what
every other parity test runs; this file checks mode 1 AGAINST THE ORACLE with the same cases and the same bar as
test_gpu_parity.py::test_time_step_parity_10_steps, states how far one tendency evaluation moves, and shows that the mode changes
nothing but the role kernel."""
import numpy as np
import pytest

from helpers import field_pairs, make_pair, rel_err, set_both, tanh_faces
from test_gpu_parity import TOPOS, TOPOS_XY

pytestmark = pytest.mark.gpu


@pytest.fixture
def contracted(ocn):
    ocn.set_option("arithmetic", 1)
    yield
    ocn.set_option("arithmetic", 0)


def test_option_is_validated_and_off_by_default(ocn, arch):
    grid = ocn.RectilinearGrid(arch, size=(16, 16, 16), extent=(1, 1, 1))
    model = ocn.NonhydrostaticModel(grid=grid, advection=ocn.WENO(), tracers=("T", "S"))
    assert model.get_option("arithmetic") == 0
    with pytest.raises(ocn.OcnError):
        ocn.set_option("arithmetic", 2)
    assert model.get_option("arithmetic") == 0


@pytest.mark.parametrize("topology", [TOPOS[0], TOPOS[1]])
def test_one_tendency_evaluation_moves_by_round_off_only(ocn, oracle, arch, topology, contracted):
    """random O(1) data, identical inputs on both sides: every tendency within 2e-14 of max|G| of the oracle's (a flux is perturbed by
    a few 2^-53 relative; the divergence of fluxes of size |u c| A divided by the cell volume amplifies that by ~ N), and NOT identical
    -- the mode really ran"""
    size = (16, 12, 10)
    z = tanh_faces(size[2]) if topology[2] == "Bounded" else None
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=z)
    set_both(ocn, m_gpu, m_cpu, seed=11, enforce_incompressibility=False)
    m_gpu.set_option("tendency_impl", 2)
    ocn.update_state(m_gpu, True)
    m_cpu.update_state(True)
    differs = False
    for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
        G_gpu, G_cpu = m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)
        err = np.abs(G_gpu - G_cpu).max() / np.abs(G_cpu).max()
        assert err < 2e-14, (n, err)
        differs = differs or not np.array_equal(G_gpu, G_cpu)
    assert differs
    # the per-field kernels ignore the option: still bit-identical
    m_gpu.set_option("tendency_impl", 0)
    ocn.update_state(m_gpu, True)
    for n, cn in zip(m_gpu.fields().keys(), ["u", "v", "w", "c0", "c1"]):
        assert np.array_equal(m_gpu.tendency(n).parent(), m_cpu.field("G" + cn)), n


@pytest.mark.parametrize("topology,stretched", [(TOPOS[0], False), (TOPOS[1], True), (TOPOS[1], False),
                                                (TOPOS_XY[0], False), (TOPOS_XY[1], False), (TOPOS_XY[1], True),
                                                (TOPOS_XY[2], False), (TOPOS_XY[3], False), (TOPOS_XY[4], True)])
def test_ten_steps_stay_within_1e12_of_the_oracle(ocn, oracle, arch, topology, stretched, contracted, record_property):
    """the nine cases of test_time_step_parity_10_steps with the contracted flux: same bar (the role kernel serves the x, y Periodic
    ones; the others take the per-field kernels, which ignore the option)"""
    size = (16, 16, 16)
    z = tanh_faces(size[2]) if stretched else None
    g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, oracle, arch, size, topology, z=z)
    set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
    dt = 0.1 * g_gpu.Δxᶜᵃᵃ / 0.6
    for _ in range(10):
        ocn.time_step(m_gpu, dt)
        m_cpu.time_step(dt)
    worst = 0.0
    for name, a, b in field_pairs(m_gpu, m_cpu):
        e = rel_err(a[3:-3, 3:-3, 3:-3], b[3:-3, 3:-3, 3:-3])
        worst = max(worst, e)
        assert e < 1e-12, (name, e)
    record_property("max_rel_err_10_steps", worst)
    print(f"[arithmetic=1] {topology} stretched={stretched}: max rel err after 10 steps {worst:.2e}")
    assert ocn.max_abs_divergence(m_gpu) < 5e-8


def test_drift_on_the_survey_state_with_the_offset_tracer(ocn, arch, contracted):
    """the S = 35 + sin cos state of SURVEY.md 8(d) (tests/offset_tracer.py): mode 1 against mode 0 ON THE DEVICE at 64^3 -- u, v, w, T
    within 1e-12; S inside the bound every round-off-equivalent pair of evaluations obeys (and not better: the mode is one more such
    pair)"""
    from offset_tracer import offset_tracer_bound, survey_state
    size, nsteps = (64, 64, 64), 3
    outs = []
    for mode in (1, 0):
        ocn.set_option("arithmetic", mode)
        grid = ocn.RectilinearGrid(arch, size=size, extent=(1, 1, 1))
        model = ocn.NonhydrostaticModel(grid=grid, advection=ocn.WENO(), tracers=("T", "S"))
        ocn.set_model(model, **survey_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}))
        dt = 0.1 / 64 / 0.6
        for _ in range(nsteps):
            ocn.time_step(model, dt)
        outs.append({n: f.interior() for n, f in model.fields().items()})
        model.close()
    errs = {n: float(np.abs(outs[0][n] - outs[1][n]).max() / np.abs(outs[1][n]).max()) for n in outs[0]}
    print(f"[arithmetic=1 vs 0] 64^3 survey state, {nsteps} steps: " + " ".join(f"{n}:{e:.2e}" for n, e in errs.items()))
    for n in ("u", "v", "w", "T"):
        assert errs[n] < 1e-12, errs
    assert errs["S"] <= offset_tracer_bound(size, nsteps), errs
