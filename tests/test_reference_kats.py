"""Known answers the reference itself holds in docstrings and tests that are not field data: the
`permute_index` / `unpermute_index` tables (src/Solvers/index_permutations.jl:5-35), `load_weno_stencil` (src/Advection/
weno_interpolants.jl:366-374), the `WENO()` composition (weno_reconstruction.jl:53-75), and the halo / advection-order expectations of
test/test_nonhydrostatic_models.jl:40-97 (adapt_advection_order + inflate_grid_halo_size). CPU side: the oracle's restatements and the
host mirror. The HIP side of the same answers is in tests/test_gpu_parity.py (test_permutation_tables_on_device,
test_adapted_advection_order_*).

Parity stays UNPINNED against Julia output: these pin index work and scheme selection, not WENO-5 / RK3 field values."""
import ctypes as C

import numpy as np
import pytest

# the tables of the docstrings, index_permutations.jl:8-13 and :26-31
PERMUTE = {8: [1, 8, 2, 7, 3, 6, 4, 5], 9: [1, 9, 2, 8, 3, 7, 4, 6, 5]}
UNPERMUTE = {8: [1, 3, 5, 7, 8, 6, 4, 2], 9: [1, 3, 5, 7, 9, 8, 6, 4, 2]}


@pytest.mark.parametrize("N", [8, 9])
def test_permute_and_unpermute_index_tables(oracle, N):
    L = oracle.lib()
    assert [L.oro_permute_index(i, N) for i in range(1, N + 1)] == PERMUTE[N]
    assert [L.oro_unpermute_index(i, N) for i in range(1, N + 1)] == UNPERMUTE[N]


@pytest.mark.parametrize("N", [1, 2, 3, 7, 8, 9, 16, 27, 128])
def test_permutations_are_mutually_inverse(oracle, N):
    """unpermute_indices!(permute_indices!(a)) == a: dst[permute_index(i)] = src[i] followed by dst[unpermute_index(i)] = src[i]"""
    L = oracle.lib()
    a = np.arange(1, N + 1)
    b = np.empty_like(a)
    for i in range(1, N + 1):
        b[L.oro_permute_index(i, N) - 1] = a[i - 1]
    c = np.empty_like(a)
    for i in range(1, N + 1):
        c[L.oro_unpermute_index(i, N) - 1] = b[i - 1]
    assert np.array_equal(c, a)
    assert sorted(L.oro_permute_index(i, N) for i in range(1, N + 1)) == list(a)


@pytest.mark.parametrize("N", [2, 3, 8, 9, 16, 27, 64])
def test_gpu_route_cosine_transform_equals_the_cpu_route(oracle, N):
    """the reference has two routes to the same cosine transforms: FFTW REDFT10 / REDFT01 x 1/2N on the CPU (discrete_transforms.jl:
    20-34) and permute + FFT + twiddle on the GPU (:108-175, Makhoul 1980). The oracle's Poisson solver uses the first, the HIP
    library the second: they must agree, and backward(forward(x)) = x."""
    L = oracle.lib()
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(N)
    x = rng.standard_normal(N)
    for backward in (0, 1):
        a, b = x.copy(), x.copy()
        L.oro_dct_direct(a.ctypes.data_as(dp), N, backward)
        L.oro_dct_makhoul(b.ctypes.data_as(dp), N, backward)
        assert np.max(np.abs(a - b)) < 1e-13 * max(1.0, np.max(np.abs(a))), (N, backward)
    y = x.copy()
    L.oro_dct_makhoul(y.ctypes.data_as(dp), N, 0)
    L.oro_dct_makhoul(y.ctypes.data_as(dp), N, 1)
    assert np.max(np.abs(y - x)) < 1e-13


@pytest.mark.parametrize("sign", [+1.0, -1.0])
def test_load_weno_stencil_footprint(oracle, sign):
    """load_weno_stencil(3, :x) = (ψ[i-3], ..., ψ[i+2]) (weno_interpolants.jl:366-374): the face value at i reads exactly these six
    cells -- the left-biased reconstruction (ũ > 0) the first five, the right-biased one the last five. A unit impulse in c on the
    reference's (Nx, 1, 1) Flat grids therefore reaches the tracer tendency of cells i0-2 .. i0+3 (ũ > 0) or i0-3 .. i0+2 (ũ < 0)
    and no other."""
    O = oracle
    N, i0 = 24, 12                                             # 1-based cell of the impulse
    g = O.Grid((N, 1, 1), topology=(O.PERIODIC, O.FLAT, O.FLAT))
    H = g.H[0]
    u = g.zeros(O.LOC["u"]); v = g.zeros(O.LOC["v"]); w = g.zeros(O.LOC["w"]); c = g.zeros(O.LOC["c"])
    u[...] = sign
    c[i0 - 1 + H, 0, 0] = 1.0
    g.fill_halo_regions(c, O.LOC["c"])
    G = g.zeros(O.LOC["c"])
    g.compute_G("c", u, v, w, G, c=c)
    Gi = g.interior_cells(G)[:, 0, 0]
    touched = set(np.nonzero(Gi)[0] + 1)
    expect = set(range(i0 - 2, i0 + 4)) if sign > 0 else set(range(i0 - 3, i0 + 3))
    assert touched <= expect
    assert min(expect) in touched and max(expect) in touched


def test_weno_composition_docstring(ocn_host):
    """weno_reconstruction.jl:53-75 (jldoctest)"""
    ocn = ocn_host
    assert repr(ocn.WENO()) == ("WENO{3, Float64, Float32}(order=5)\n├── buffer_scheme: WENO{2, Float64, Float32}(order=3)\n"
                                "└── advection_velocity_scheme: Centered(order=4)")
    assert repr(ocn.WENO(order=9)) == ("WENO{5, Float64, Float32}(order=9)\n├── buffer_scheme: WENO{4, Float64, Float32}(order=7)\n"
                                       "└── advection_velocity_scheme: Centered(order=8)")
    assert repr(ocn.WENO(order=9, bounds=(0, 1))) == ("WENO{5, Float64, Float32}(order=9)\n├── bounds: (0, 1)\n"
                                                      "├── buffer_scheme: WENO{4, Float64, Float32}(order=7)\n"
                                                      "└── advection_velocity_scheme: Centered(order=8)")
    # WENO(order=1) is UpwindBiased(order=1) (:81-83); the cascade ends there
    w = ocn.WENO()
    assert isinstance(w.buffer_scheme.buffer_scheme, ocn.UpwindBiased) and w.buffer_scheme.buffer_scheme.buffer == 1
    assert w.buffer_scheme.advecting_velocity_scheme == ocn.Centered(order=2)
    assert w.buffer_scheme.buffer_scheme.advecting_velocity_scheme == ocn.Centered(order=2)
    with pytest.raises(ValueError):
        ocn.WENO(order=4)
    with pytest.raises(ValueError):
        ocn.Centered(order=3)
    with pytest.raises(ValueError):
        ocn.UpwindBiased(order=2)
    # required_halo_size_x(Centered(order=4)) == 2 (Grids/automatic_halo_sizing.jl:10-18, jldoctest)
    assert ocn.required_halo_size_x(ocn.Centered(order=4)) == 2
    assert ocn.required_halo_size_y(ocn.Centered(order=4)) == 2
    assert ocn.required_halo_size_z(ocn.Centered(order=4)) == 2


@pytest.fixture()
def ocn_host():
    """the host mirror without a device: grids are metadata until a kernel needs their handle"""
    import oldoceananigans_jl_amd as ocn
    return ocn


def _model_grid_and_advection(ocn, grid, advection, closure=None):
    """what the NonhydrostaticModel constructor settles before it allocates anything (nonhydrostatic_model.jl:176-184)"""
    advection = ocn.adapt_advection_order(advection, grid)
    required = ocn.inflate_halo_size(*grid.halo_size, grid, advection, closure)
    if any(u < r for u, r in zip(grid.halo_size, required)):
        grid = ocn.with_halo(required, grid)
    return grid, advection


def test_adjustment_of_halos_in_the_model_constructor(ocn_host):
    """test/test_nonhydrostatic_models.jl:40-70"""
    ocn = ocn_host
    minimal_grid = ocn.RectilinearGrid(None, size=(4, 4, 4), extent=(1, 2, 3), halo=(1, 1, 1))
    funny_grid = ocn.RectilinearGrid(None, size=(4, 4, 4), extent=(1, 2, 3), halo=(1, 3, 4))
    g, _ = _model_grid_and_advection(ocn, minimal_grid, None)             # the reference's default advection is Centered(order=2)
    assert g.halo_size == (1, 1, 1)
    g, _ = _model_grid_and_advection(ocn, funny_grid, ocn.Centered())
    assert g.halo_size == (1, 3, 4)
    for scheme in (ocn.Centered(order=4), ocn.UpwindBiased(order=3)):
        assert _model_grid_and_advection(ocn, minimal_grid, scheme)[0].halo_size == (2, 2, 2)
        assert _model_grid_and_advection(ocn, funny_grid, scheme)[0].halo_size == (2, 3, 4)
    for scheme in (ocn.WENO(), ocn.UpwindBiased(order=5)):
        assert _model_grid_and_advection(ocn, minimal_grid, scheme)[0].halo_size == (3, 3, 3)
        assert _model_grid_and_advection(ocn, funny_grid, scheme)[0].halo_size == (3, 3, 4)
    # the inflated grid is the same grid otherwise
    g = _model_grid_and_advection(ocn, funny_grid, ocn.WENO())[0]
    assert (g.size, g.Lx, g.Ly, g.Lz, g.topology) == (funny_grid.size, 1.0, 2.0, 3.0, funny_grid.topology)
    # default halo: min(3, size) per direction (Grids/input_validation.jl:71-77); halo <= size in x and y (:86-92)
    assert ocn.RectilinearGrid(None, size=(4, 2, 4), extent=(1, 2, 3)).halo_size == (3, 2, 3)
    with pytest.raises(ValueError):
        ocn.RectilinearGrid(None, size=(4, 2, 4), extent=(1, 2, 3), halo=(3, 3, 3))


def test_adjustment_of_advection_schemes_in_the_model_constructor(ocn_host):
    """test/test_nonhydrostatic_models.jl:72-91: small_grid = (4, 2, 4), halo (1, 1, 1)"""
    ocn = ocn_host
    small_grid = ocn.RectilinearGrid(None, size=(4, 2, 4), extent=(1, 2, 3), halo=(1, 1, 1))
    for scheme, expected in ((ocn.WENO(), (3, 2, 3)), (ocn.UpwindBiased(order=9), (4, 2, 4)), (ocn.Centered(order=10), (4, 2, 4))):
        grid, advection = _model_grid_and_advection(ocn, small_grid, scheme)
        assert isinstance(advection, ocn.FluxFormAdvection)
        assert (ocn.required_halo_size_x(advection), ocn.required_halo_size_y(advection),
                ocn.required_halo_size_z(advection)) == expected
        assert grid.halo_size == expected
        assert type(advection.y) is type(scheme)                       # the family is kept, the order is 2N - 1 (2N for Centered)
    # nothing changes on a grid that is large enough: the scheme itself comes back (adapt_advection_order.jl:49)
    big = ocn.RectilinearGrid(None, size=(8, 8, 8), extent=(1, 1, 1))
    w = ocn.WENO()
    assert ocn.adapt_advection_order(w, big) is w
    # Flat directions are not adapted (:62-63)
    flat = ocn.RectilinearGrid(None, size=(8, 8), extent=(1, 1), topology=(ocn.Periodic, ocn.Flat, ocn.Bounded))
    assert ocn.adapt_advection_order(w, flat) is w
    # the oracle's restatement of the same rule, and what the oracle's grid derives from WENO(order=5)
    from oracle import oracle as O
    L = O.lib()
    for family, B in ((0, 5), (1, 5), (2, 3)):
        for N in range(1, 8):
            assert L.oro_adapt_advection_order(family, B, N, O.PERIODIC) == min(B, N)
            assert L.oro_adapt_advection_order(family, B, N, O.FLAT) == B
    assert O.Grid((4, 2, 4)).B == (3, 2, 3) and O.Grid((4, 2, 4)).H == (3, 2, 3)


def test_reduced_order_direction_uses_weno3_and_centered2(oracle):
    """a direction with N = 2 carries WENO(order=3): tracer flux Ay v cᴿ with cᴿ from the four-point stencil ψ[j-2 .. j+1]
    (weno_interpolants.jl:366-374, buffer 2), and the fluxes that point along it interpolate their advecting transport with
    Centered(order=2) (weno_reconstruction.jl:87). Hand evaluation with the oracle's WENO{2} function."""
    O = oracle
    L = O.lib()
    dp = C.POINTER(C.c_double)
    g = O.Grid((6, 2, 5))
    assert g.B == (3, 2, 3)
    rng = np.random.default_rng(5)
    u = g.zeros(O.LOC["u"]); v = g.zeros(O.LOC["v"]); w = g.zeros(O.LOC["w"]); c = g.zeros(O.LOC["c"])
    Hx, Hy, Hz = g.H
    vi = rng.standard_normal((6, 2, 5))
    ci = rng.standard_normal((6, 2, 5))
    v[Hx:-Hx, Hy:-Hy, Hz:-Hz] = vi
    c[Hx:-Hx, Hy:-Hy, Hz:-Hz] = ci
    for a, loc in ((v, "v"), (c, "c")):
        g.fill_halo_regions(a, O.LOC[loc])
    G = g.zeros(O.LOC["c"])
    g.compute_G("c", u, v, w, G, c=c)
    dx, dy, dz = g.dc[0][0], g.dc[1][0], g.dc[2][0]

    def flux(i, j, k):                                                    # 0-based parent indices of the face
        vt = v[i, j, k]
        S = np.ascontiguousarray(c[i, j - 2:j + 2, k])
        cr = L.oro_weno3_biased(S.ctypes.data_as(dp), int(vt > 0))
        return (dx * dz) * vt * cr

    for i in range(6):
        for j in range(2):
            for k in range(5):
                I, J, K = i + Hx, j + Hy, k + Hz
                div = (1.0 / ((dx * dy) * dz)) * ((0.0 + (flux(I, J + 1, K) - flux(I, J, K))) + 0.0)
                assert G[I, J, K] == -div + 0.0, (i, j, k)
    # momentum: u = u(y) only, w = 0, v random -> the x- and z-flux differences of Gu vanish exactly and the only flux left is
    # Vu = ṽ uᴿ with ṽ the TWO-point average of Ay v along x (Centered(order=2): the flux points along y, whose scheme is WENO{2})
    from fractions import Fraction

    def fma(a, b, cc):                                                    # one rounding, like the hardware instruction
        return float(Fraction(a) * Fraction(b) + Fraction(cc))

    u[...] = 0.0
    u[Hx:-Hx, Hy:-Hy, Hz:-Hz] = rng.standard_normal((1, 2, 1))
    g.fill_halo_regions(u, O.LOC["u"])
    Gu = g.zeros(O.LOC["u"])
    g.compute_G("u", u, v, w, Gu)

    def Vu(i, j, k):
        q1, q2 = (dx * dz) * v[i - 1, j, k], (dx * dz) * v[i, j, k]
        vt = fma(0.5, q2, 0.5 * q1)
        S = np.ascontiguousarray(u[i, j - 2:j + 2, k])
        return vt * L.oro_weno3_biased(S.ctypes.data_as(dp), int(vt > 0))

    for i in range(6):
        for j in range(2):
            for k in range(5):
                I, J, K = i + Hx, j + Hy, k + Hz
                div = (1.0 / ((dx * dy) * dz)) * ((0.0 + (Vu(I, J + 1, K) - Vu(I, J, K))) + 0.0)
                assert Gu[I, J, K] == -div + 0.0, (i, j, k)


@pytest.mark.parametrize("perm", [(0, 1, 2), (1, 0, 2), (2, 1, 0)])
def test_adapted_model_time_steps_and_is_direction_symmetric(oracle, perm):
    """NonhydrostaticModel(grid = (8, 2, 6)-like, advection = WENO()) as the reference builds it (adapted y scheme, halo (3, 2, 3)):
    stays finite, divergence-free (test/test_time_stepping.jl:124-160) and conserves the tracer mean; the same physical problem
    with the short direction along x, y or z gives the same answer (the convergence tests' cx ≈ cy ≈ cz symmetry check,
    validation/convergence_tests/one_dimensional_advection_schemes.jl:108-118, here for the adapted scheme)."""
    O = oracle
    base = (8, 2, 6)
    size = tuple(base[perm.index(d)] for d in range(3))                  # direction d of this run is base direction perm.index(d)

    def run(size, axes):
        g = O.Grid(size)
        m = O.Model(g, 1)
        rng = np.random.default_rng(3)
        fields = {n: rng.standard_normal(base) for n in ("u", "v", "w", "c0")}
        vel = [fields["u"], fields["v"], fields["w"]]
        # base velocity component b becomes component axes[b] of this run, arrays transposed accordingly
        args = {}
        for b in range(3):
            args["uvw"[axes[b]]] = np.ascontiguousarray(np.moveaxis(vel[b], (0, 1, 2), axes))
        args["c0"] = np.ascontiguousarray(np.moveaxis(fields["c0"], (0, 1, 2), axes))
        m.set(**args)
        mean0 = g.interior_cells(m.field("c0")).mean()
        for _ in range(3):
            m.time_step(1e-3)
        assert m.max_abs_divergence() < 5e-8
        c = g.interior_cells(m.field("c0"))
        assert np.all(np.isfinite(c)) and abs(c.mean() - mean0) < 1e-13
        # copies: the oracle's arrays die with the model
        return (np.moveaxis(c, axes, (0, 1, 2)).copy(),
                [np.moveaxis(g.interior_cells(m.field("uvw"[axes[b]])), axes, (0, 1, 2)).copy() for b in range(3)])

    c_ref, vel_ref = run(base, (0, 1, 2))
    c, vel = run(size, perm)
    assert np.max(np.abs(c - c_ref)) < 1e-12 * np.max(np.abs(c_ref))
    for a, b in zip(vel, vel_ref):
        assert np.max(np.abs(a - b)) < 1e-11 * np.max(np.abs(b))


# ---------------------------------------------------------------------------------------------------------------------
# array-valued boundary conditions: getbc(condition::AbstractArray, i, j, grid, args...) = condition[i, j]
# (src/BoundaryConditions/boundary_condition.jl:164; used by compute_flux_bcs.jl:114-163 and fill_halo_regions_value_gradient.jl:7-119)
# ---------------------------------------------------------------------------------------------------------------------
def test_array_valued_conditions_reduce_to_numbers_and_index_the_tangential_point(oracle):
    O = oracle
    N = (6, 5, 4)
    g = O.Grid(N, topology=(O.BOUNDED, O.BOUNDED, O.BOUNDED), z=(-1.0, 0.0))
    rng = np.random.default_rng(8)
    loc = O.LOC["c"]
    shape = {"west": (N[1], N[2]), "east": (N[1], N[2]), "south": (N[0], N[2]), "north": (N[0], N[2]), "bottom": (N[0], N[1]), "top": (N[0], N[1])}
    kinds = {"west": "value", "east": "gradient", "south": "gradient", "north": "value", "bottom": "value", "top": "gradient"}
    # 1. an array filled with one number gives the bits of the number
    c0 = np.asfortranarray(rng.standard_normal(g.parent_size(loc)))
    a, b = c0.copy(order="F"), c0.copy(order="F")
    g.fill_halo_regions(a, loc, True, bcs={s: (k, 0.37) for s, k in kinds.items()})
    g.fill_halo_regions(b, loc, True, bcs={s: (k, np.full(shape[s], 0.37)) for s, k in kinds.items()})
    assert np.array_equal(a, b)
    # 2. a varying array: the halo cell behind boundary point (i, j) follows condition[i, j]
    arrs = {s: rng.standard_normal(shape[s]) for s in kinds}
    c = c0.copy(order="F")
    g.fill_halo_regions(c, loc, True, bcs={s: (kinds[s], arrs[s]) for s in kinds})
    H = 3
    I = (slice(H, H + N[0]), slice(H, H + N[1]), slice(H, H + N[2]))
    dx, dy, dz = 1.0 / N[0], 1.0 / N[1], 1.0 / N[2]
    assert np.allclose((c[H - 1, I[1], I[2]] + c[H, I[1], I[2]]) / 2, arrs["west"], rtol=0, atol=1e-13)
    assert np.allclose((c[H + N[0], I[1], I[2]] - c[H + N[0] - 1, I[1], I[2]]) / dx, arrs["east"], rtol=0, atol=1e-12)
    assert np.allclose((c[I[0], H, I[2]] - c[I[0], H - 1, I[2]]) / dy, arrs["south"], rtol=0, atol=1e-12)
    assert np.allclose((c[I[0], H + N[1], I[2]] + c[I[0], H + N[1] - 1, I[2]]) / 2, arrs["north"], rtol=0, atol=1e-13)
    assert np.allclose((c[I[0], I[1], H - 1] + c[I[0], I[1], H]) / 2, arrs["bottom"], rtol=0, atol=1e-13)
    assert np.allclose((c[I[0], I[1], H + N[2]] - c[I[0], I[1], H + N[2] - 1]) / dz, arrs["top"], rtol=0, atol=1e-12)
    # 3. Flux arrays: G[1] += flux[i, j] A / V, G[N] -= flux[i, j] A / V (compute_flux_bcs.jl:57-163)
    G = g.zeros(loc)
    fl = {"west": rng.standard_normal(shape["west"]), "top": rng.standard_normal(shape["top"])}
    g.compute_flux_bcs(G, loc, {s: ("flux", fl[s]) for s in fl})
    Gi = g.interior_cells(G)
    expect = np.zeros(N)
    expect[0, :, :] += fl["west"] * (dy * dz) / ((dx * dy) * dz)
    expect[:, :, -1] -= fl["top"] * (dx * dy) / ((dx * dy) * dz)
    assert np.allclose(Gi, expect, rtol=1e-15, atol=0)
    # Open arrays on the wall-normal component
    w = g.zeros(O.LOC["w"])
    top = rng.standard_normal(shape["top"])
    g.fill_halo_regions(w, O.LOC["w"], True, bcs={"top": ("open", top)})
    assert np.array_equal(w[H:H + N[0], H:H + N[1], H + N[2]], top)


def test_array_flux_condition_budget(oracle):
    """the flux budget of test/test_boundary_conditions_integration.jl:28-52 with a spatially varying flux array: after one RK3 step with
    Δt = 1 from rest, mean(c) = -mean(flux_top) t / Lz (a tracer at rest is only moved by its boundary flux)"""
    O = oracle
    Lz = 0.5
    g = O.Grid((4, 6, 4), topology=(O.PERIODIC, O.PERIODIC, O.BOUNDED), z=(0.0, Lz))
    m = O.Model(g, 1)
    rng = np.random.default_rng(1)
    flux = rng.standard_normal((4, 6))
    m.set_bc("c0", "top", "flux", flux)
    m.set(**{n: 0.0 for n in m.names()})
    m.time_step(1.0)
    c = g.interior_cells(m.field("c0"))
    assert abs(c.mean() - (-flux.mean() * m.time / Lz)) < 1e-13
    # ... and column by column (nothing mixes the columns of a fluid at rest): the top cell of column (i, j) took -flux[i, j] t / Δz
    assert np.allclose(c[:, :, -1], -flux * m.time / (Lz / 4), rtol=1e-13, atol=0)


# ---------------------------------------------------------------------------------------------------------------------
# grid generation: the numbers the reference's docstrings print (src/Grids/rectilinear_grid.jl:158-262 jldoctests,
# nodes_and_spacings.jl:150-196, automatic_halo_sizing.jl:10-57) -- reference-held known answers for row a1
# ---------------------------------------------------------------------------------------------------------------------
def _six(v):
    """the six significant digits of the reference's grid summaries"""
    return float(f"{float(v):.6g}")


def test_rectilinear_grid_docstring_examples(ocn_host, oracle):
    ocn = ocn_host
    PPB = (ocn.Periodic, ocn.Periodic, ocn.Bounded)
    # "32×32×32 RectilinearGrid{Float64, Periodic, Periodic, Bounded} on CPU with 3×3×3 halo": Δx=0.03125, Δy=0.0625, Δz=0.09375, z ∈ [-3.0, 0.0]
    g = ocn.RectilinearGrid(None, size=(32, 32, 32), extent=(1, 2, 3), topology=PPB)
    assert (g.Δxᶜᵃᵃ, g.Δyᵃᶜᵃ, g.Δzᵃᵃᶜ[3]) == (0.03125, 0.0625, 0.09375) and g.halo_size == (3, 3, 3)
    assert (g.x0, g.x0 + g.Lx, g.y0 + g.Ly, g.z0, g.z0 + g.Lz) == (0.0, 1.0, 2.0, -3.0, 0.0)
    assert repr(g).startswith("32×32×32 RectilinearGrid{Float64, Periodic, Periodic, Bounded}")
    # size=(32, 32), extent=(2π, 4π), (Periodic, Periodic, Flat): "32×32×1 ... with 3×3×0 halo", Δx=0.19635, Δy=0.392699
    g = ocn.RectilinearGrid(None, size=(32, 32), extent=(2 * np.pi, 4 * np.pi), topology=(ocn.Periodic, ocn.Periodic, ocn.Flat))
    assert g.size == (32, 32, 1) and g.halo_size == (3, 3, 0) and (_six(g.Δxᶜᵃᵃ), _six(g.Δyᵃᶜᵃ)) == (0.19635, 0.392699)
    assert (_six(g.x0 + g.Lx), _six(g.y0 + g.Ly)) == (6.28319, 12.5664)
    # size=256, z=(-128, 0), (Flat, Flat, Bounded): "1×1×256 ... with 0×0×3 halo", Δz=0.5
    g = ocn.RectilinearGrid(None, size=256, z=(-128, 0), topology=(ocn.Flat, ocn.Flat, ocn.Bounded))
    assert g.size == (1, 1, 256) and g.halo_size == (0, 0, 3) and g.Δzᵃᵃᶜ[3] == 0.5 and (g.z0, g.z0 + g.Lz) == (-128.0, 0.0)
    # hyperbolically spaced faces, σ = 1.1, Nz = 24, Lz = 32: "z ∈ [-32.0, -0.0] variably spaced with min(Δz)=0.682695, max(Δz)=1.83091"
    sigma, Nz, Lz = 1.1, 24, 32

    def hyperbolically_spaced_faces(k):
        return -Lz * (1 - np.tanh(sigma * (k - 1) / Nz) / np.tanh(sigma))
    g = ocn.RectilinearGrid(None, size=(32, 32, Nz), x=(0, 64), y=(0, 64), z=hyperbolically_spaced_faces, topology=PPB)
    dz = np.asarray(g.Δzᵃᵃᶜ[3:3 + Nz])
    assert (g.Δxᶜᵃᵃ, g.Δyᵃᶜᵃ) == (2.0, 2.0) and (_six(dz.min()), _six(dz.max())) == (0.682695, 1.83091)
    assert g.zᵃᵃᶠ[3] == -32.0 and g.zᵃᵃᶠ[3 + Nz] == 0.0 and np.signbit(g.zᵃᵃᶠ[3 + Nz])          # the summary prints "-0.0"
    # the oracle's grid twin generates the same spacings
    faces = np.array([hyperbolically_spaced_faces(k) for k in range(1, Nz + 2)])
    go = oracle.Grid((32, 32, Nz), topology=(0, 0, 1), x=(0.0, 64.0), y=(0.0, 64.0), z=faces)
    dzo = np.asarray(go.dc[2][3:3 + Nz])
    assert np.array_equal(dzo, dz) and (go.dc[0][0], go.dc[1][0]) == (2.0, 2.0)


def test_minimum_spacing_and_required_halo_docstrings(ocn_host):
    """nodes_and_spacings.jl:150-196: size (2, 4, 8), extent (1, 1, 1): minimum x / y / z spacing 0.5, 0.25, 0.125;
    automatic_halo_sizing.jl:10-57: required_halo_size_x / y / z(Centered(order=4)) = 2"""
    ocn = ocn_host
    g = ocn.RectilinearGrid(None, size=(2, 4, 8), extent=(1, 1, 1), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    assert (g.Δxᶜᵃᵃ, g.Δyᵃᶜᵃ, float(np.min(g.Δzᵃᵃᶜ[g.Hz:g.Hz + 8]))) == (0.5, 0.25, 0.125)
    from oldoceananigans_jl_amd.advection import required_halo_size_x, required_halo_size_y, required_halo_size_z
    c4 = ocn.Centered(order=4)
    assert (required_halo_size_x(c4), required_halo_size_y(c4), required_halo_size_z(c4)) == (2, 2, 2)


def test_closure_and_buoyancy_docstring_defaults(ocn_host):
    """scalar_diffusivity.jl:75-80, anisotropic_minimum_dissipation.jl:81-103 (jldoctests: the constants the kernels receive),
    seawater_buoyancy.jl:80-83 (gravitational_acceleration: 9.80665), linear_equation_of_state.jl defaults (1.67e-4, 7.80e-4)"""
    ocn = ocn_host
    assert repr(ocn.ScalarDiffusivity(ν=1000, κ=2000)) == "ScalarDiffusivity{ExplicitTimeDiscretization}(ν=1000.0, κ=2000.0)"
    a = ocn.AnisotropicMinimumDissipation(C=1 / 2)
    assert (a.Cν, a.Cκ, a.Cb) == (0.5, 0.5, None)
    assert ocn.AnisotropicMinimumDissipation().Cν == 0.3333333333333333          # "Cν: 0.3333333333333333" in the second example
    b = ocn.SeawaterBuoyancy()
    assert b.gravitational_acceleration == 9.80665
    assert (b.equation_of_state.thermal_expansion, b.equation_of_state.haline_contraction) == (1.67e-4, 7.80e-4)


def test_node_coordinates_are_julia_ranges(ocn_host):
    """grid_generation.jl:104-125: face and centre coordinates are `range(FT(F₋), FT(F₊), length = TF)` with F₋ = c₁ - H Δ evaluated in
    BigFloat -- Julia's twice-precision StepRangeLen, not c₁ + (i - 1) Δ. The reference's grid summaries print the first face, which
    pins the restatement (oldoceananigans.jl_amd/grids.py: julia_range): rectilinear_grid.jl:193-195 (halo 3) and docs/src/grids.md:311-313
    (halo 7)."""
    ocn = ocn_host
    PPF = (ocn.Periodic, ocn.Periodic, ocn.Flat)
    g = ocn.RectilinearGrid(None, size=(32, 32), extent=(2 * np.pi, 4 * np.pi), topology=PPF)
    x, y, _ = g.nodes((ocn.Face, ocn.Face, ocn.Center))
    assert (f"{x.ravel()[0]:.6g}", f"{y.ravel()[0]:.6g}") == ("3.60072e-17", "7.20145e-17")
    g = ocn.RectilinearGrid(None, size=(32, 16), halo=(7, 7), x=(0, 2 * np.pi), y=(0, np.pi), topology=PPF)
    x, y, _ = g.nodes((ocn.Face, ocn.Face, ocn.Center))
    assert (f"{x.ravel()[0]:.6g}", f"{y.ravel()[0]:.6g}") == ("-6.90805e-17", "-1.07194e-16")
    assert _six(g.Δxᶜᵃᵃ) == 0.19635 and _six(g.Δyᵃᶜᵃ) == 0.19635 and g.halo_size == (7, 7, 0)
    # exactly representable spacings give exact nodes: docs/src/grids.md:26-36 (16 x 8 x 4 on 64 x 32 x 8: Δ = 4, 4, 2)
    g = ocn.RectilinearGrid(None, size=(16, 8, 4), x=(0, 64), y=(0, 32), z=(0, 8), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    xc, yc, zf = g.nodes((ocn.Center, ocn.Center, ocn.Face))
    assert np.array_equal(xc.ravel(), 2.0 + 4.0 * np.arange(16)) and np.array_equal(yc.ravel(), 2.0 + 4.0 * np.arange(8))
    assert np.array_equal(zf.ravel(), [0.0, 2.0, 4.0, 6.0, 8.0]) and (g.Δxᶜᵃᵃ, g.Δyᵃᶜᵃ, g.Δzᵃᵃᶜ[3]) == (4.0, 4.0, 2.0)
    # docs/src/grids.md:54-66: z faces [0, 1, 3, 6, 10] on (Periodic, Flat, Bounded): min(Δz) = 1.0, max(Δz) = 4.0, "10×1×4 ... 3×0×3 halo"
    g = ocn.RectilinearGrid(None, size=(10, 4), x=(0, 20), z=[0, 1, 3, 6, 10], topology=(ocn.Periodic, ocn.Flat, ocn.Bounded))
    dz = np.asarray(g.Δzᵃᵃᶜ[3:7])
    assert g.size == (10, 1, 4) and g.halo_size == (3, 0, 3) and (dz.min(), dz.max(), g.Δxᶜᵃᵃ) == (1.0, 4.0, 2.0)
    # docs/src/grids.md:342-366: Chebychev-spaced z faces, Nz = 32, Lz = 1e3: min(Δz)=2.40764, max(Δz)=49.0086, z ∈ [-1000.0, -0.0]
    Nz, Lz = 32, 1e3
    g = ocn.RectilinearGrid(None, size=(64, 64, Nz), x=(0, 1e4), y=(0, 1e4), z=lambda k: -Lz * (1 + np.cos(np.pi * (k - 1) / Nz)) / 2,
                            topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    dz = np.asarray(g.Δzᵃᵃᶜ[3:3 + Nz])
    assert (_six(dz.min()), _six(dz.max()), g.Δxᶜᵃᵃ) == (2.40764, 49.0086, 156.25) and g.zᵃᵃᶠ[3] == -1000.0
    # the range's end points come from the USER's interval in BigFloat (c₁, c₂ = BigFloat.(node_interval), grid_generation.jl:104-110), not
    # from c₁ + FT(L): for x = (-0.1, 0.2) the rounded extent gives -0.1 + 0.30000000000000004 = 0.20000000000000004 != 0.2. A Julia range
    # hits its two end points exactly, and those are F₋ = c₁ - H Δ and F₊ = c₂ + (H - 1) Δ (Periodic) rounded ONCE
    from fractions import Fraction
    c1, c2, N, H = Fraction(-0.1), Fraction(0.2), 6, 3
    assert float(c1) + float(c2 - c1) != 0.2                       # the case tells the two constructions apart
    g = ocn.RectilinearGrid(None, size=(N, N, N), x=(-0.1, 0.2), y=(-0.1, 0.2), z=(-0.1, 0.2))
    D = (c2 - c1) / N
    for faces, centres in ((g.xᶠᵃᵃ, g.xᶜᵃᵃ), (g.yᵃᶠᵃ, g.yᵃᶜᵃ), (g.zᵃᵃᶠ, g.zᵃᵃᶜ)):
        assert faces[0] == float(c1 - H * D) and faces[-1] == float(c2 + (H - 1) * D)
        assert centres[0] == float(c1 - H * D + D / 2) and centres[-1] == float(c2 + H * D - D / 2)


def test_stretched_coordinate_full_precision_docstring_numbers(ocn_host, oracle):
    """docs/src/fields.md:18-99 (jldoctests): size (4, 5, 4), halo (1, 1, 1), z = [0, 0.1, 0.3, 0.6, 1] -- the reference prints
    zspacings at Center and Face and the halo'd znodes with all 17 digits: the stretched-coordinate generation
    (grid_generation.jl:34-95) is pinned bit for bit, in the host mirror and in the oracle's twin (the metric tables of every kernel)"""
    ocn = ocn_host
    dzc = [0.1, 0.19999999999999998, 0.3, 0.4]
    dzf = [0.1, 0.15000000000000002, 0.24999999999999994, 0.3500000000000001, 0.3999999999999999]
    zc = [-0.05, 0.05, 0.2, 0.44999999999999996, 0.8, 1.2]
    faces = [0, 0.1, 0.3, 0.6, 1]
    g = ocn.RectilinearGrid(None, size=(4, 5, 4), halo=(1, 1, 1), x=(0, 1), y=(0, 1), z=faces, topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    assert [float(v) for v in g.Δzᵃᵃᶜ[1:5]] == dzc and [float(v) for v in g.Δzᵃᵃᶠ[1:6]] == dzf and [float(v) for v in g.zᵃᵃᶜ[0:6]] == zc
    assert (g.Δxᶜᵃᵃ, g.Δyᵃᶜᵃ) == (0.25, 0.2) and [float(v) for v in g.nodes((ocn.Center, ocn.Center, ocn.Center))[2].ravel()] == zc[1:5]
    assert [float(v) for v in g.nodes((ocn.Center, ocn.Center, ocn.Face))[2].ravel()] == [0.0, 0.1, 0.3, 0.6, 1.0]       # znodes(w), fields.md:183
    # xnodes(c) = [0.125, 0.375, 0.625, 0.875], xnodes(u) = [0.0, 0.25, 0.5, 0.75] (fields.md:163-164)
    assert [float(v) for v in g.nodes((ocn.Center, ocn.Center, ocn.Center))[0].ravel()] == [0.125, 0.375, 0.625, 0.875]
    assert [float(v) for v in g.nodes((ocn.Face, ocn.Center, ocn.Center))[0].ravel()] == [0.0, 0.25, 0.5, 0.75]
    go = oracle.Grid((4, 5, 4), topology=(0, 0, 1), x=(0.0, 1.0), y=(0.0, 1.0), z=np.array(faces, dtype=float), halo=(1, 1, 1))
    assert [float(v) for v in go.dc[2][1:5]] == dzc and [float(v) for v in go.df[2][1:6]] == dzf


def test_coriolis_docstring_numbers(ocn_host):
    """docs/src/model_setup/coriolis.md:25-37: FPlane(f=1e-4) -> f=0.0001; FPlane(rotation_rate=7.292115e-5, latitude=45) -> f=0.000103126
    (2 rotation_rate sind(latitude), f_plane.jl:44; sind(45) is the correctly rounded sqrt(2)/2, one ulp above sin(45 pi / 180))"""
    ocn = ocn_host
    from oldoceananigans_jl_amd.buoyancy import sind
    assert ocn.FPlane(f=1e-4).f == 0.0001
    assert _six(ocn.FPlane(rotation_rate=7.292115e-5, latitude=45).f) == 0.000103126
    # test/test_coriolis.jl:18-26: FPlane(f=π).f ≈ π; FPlane(rotation_rate=2, latitude=30).f ≈ 2 (here: exactly, sind(30) = 1/2)
    assert ocn.FPlane(f=np.pi).f == np.pi and ocn.FPlane(rotation_rate=2, latitude=30).f == 2.0
    assert sind(45) == 0.7071067811865476 == 2 ** 0.5 / 2 and sind(30) == 0.5 and sind(90) == 1.0 and sind(-90) == -1.0 and sind(180) == 0.0


def test_field_parent_sizes_of_the_field_docstrings(ocn_host):
    """Fields/field.jl:141-148: size (2, 3, 4) -> default halo 2 x 3 x 3, a Field{Face, Face, Center} holds 6 x 9 x 10 values (indices
    -1:4, -2:6, -2:7); docs/src/fields.md: halo (1, 1, 1) on 4 x 5 x 4 -> 6 x 7 x 6; w on a Bounded z has Nz + 1 faces (grid_utils.jl:63-65)"""
    ocn = ocn_host
    PPB = (ocn.Periodic, ocn.Periodic, ocn.Bounded)
    g = ocn.RectilinearGrid(None, size=(2, 3, 4), extent=(1, 1, 1), topology=PPB)
    assert g.halo_size == (2, 3, 3) and g.total_size((ocn.Face, ocn.Face, ocn.Center)) == (6, 9, 10)
    assert g.total_size((ocn.Center, ocn.Center, ocn.Face)) == (6, 9, 11) and g.interior_size((ocn.Center, ocn.Center, ocn.Face)) == (2, 3, 5)
    g = ocn.RectilinearGrid(None, size=(4, 5, 4), halo=(1, 1, 1), x=(0, 1), y=(0, 1), z=[0, 0.1, 0.3, 0.6, 1], topology=PPB)
    assert g.total_size((ocn.Center,) * 3) == (6, 7, 6)


# ---------------------------------------------------------------------------------------------------------------------
# test/test_grids.jl:18-152,184-241,440-487: the reference's own RectilinearGrid tests on the host mirror (Float64)
# ---------------------------------------------------------------------------------------------------------------------
def test_regular_rectilinear_grid_as_the_reference_tests_it(ocn_host):
    ocn = ocn_host
    P, B = ocn.Periodic, ocn.Bounded
    pi = np.pi
    g = ocn.RectilinearGrid(None, size=(4, 6, 8), extent=(2 * pi, 4 * pi, 9 * pi), topology=(P, P, B))          # correct_size
    assert g.size == (4, 6, 8) and np.allclose((g.Lx, g.Ly, g.Lz), (2 * pi, 4 * pi, 9 * pi), rtol=1e-15)
    g = ocn.RectilinearGrid(None, size=(4, 6, 8), x=(1, 2), y=(pi, 3 * pi), z=(0, 4), topology=(P, P, B))       # correct_extent
    assert np.allclose((g.Lx, g.Ly, g.Lz), (1, 2 * pi, 4), rtol=1e-15)
    g = ocn.RectilinearGrid(None, size=(2, 3, 4), extent=(1, 1, 1), halo=(1, 1, 1), topology=(P, B, B))         # coordinate_lengths
    assert (len(g.xᶜᵃᵃ), len(g.yᵃᶜᵃ), len(g.xᶠᵃᵃ), len(g.yᵃᶠᵃ), len(g.zᵃᵃᶜ), len(g.zᵃᵃᶠ)) == (4, 5, 4, 6, 6, 7)
    g = ocn.RectilinearGrid(None, size=(4, 6, 8), extent=(2 * pi, 4 * pi, 9 * pi), halo=(1, 2, 3), topology=(P, P, B))
    assert g.halo_size == (1, 2, 3)                                                                              # correct_halo_size
    N, H, L = 4, 1, 2.0                                                                                           # correct_halo_faces
    D = L / N
    g = ocn.RectilinearGrid(None, size=(N, N, N), x=(0, L), y=(0, L), z=(0, L), halo=(H, H, H), topology=(P, B, B))
    at = lambda a, i: a[i - 1 + H]                                       # noqa: E731  OffsetArray index -> array position
    assert at(g.xᶠᵃᵃ, 0) == -H * D and at(g.yᵃᶠᵃ, 0) == -H * D and at(g.zᵃᵃᶠ, 0) == -H * D
    assert at(g.xᶠᵃᵃ, N + 1) == L and at(g.yᵃᶠᵃ, N + 2) == L + H * D and at(g.zᵃᵃᶠ, N + 2) == L + H * D        # + correct_end_faces
    L = 4.0                                                                                                        # correct_first_cells
    g = ocn.RectilinearGrid(None, size=(N, N, N), x=(0, L), y=(0, L), z=(0, L), halo=(H, H, H), topology=(P, P, B))
    assert at(g.xᶜᵃᵃ, 1) == 0.5 and at(g.yᵃᶜᵃ, 1) == 0.5 and at(g.zᵃᵃᶜ, 1) == 0.5
    g = ocn.RectilinearGrid(None, size=(8, 9, 10), extent=(1, 1, 1), halo=(1, 2, 1), topology=(B, B, B))         # ranges_have_correct_length
    assert (len(g.xᶜᵃᵃ), len(g.yᵃᶜᵃ), len(g.xᶠᵃᵃ), len(g.yᵃᶠᵃ), len(g.zᵃᵃᶜ), len(g.zᵃᵃᶠ)) == (10, 13, 11, 14, 12, 13)
    g = ocn.RectilinearGrid(None, size=(1, 1, 64), extent=(1, 1, pi / 2), halo=(1, 1, 1), topology=(P, P, B))    # no_roundoff_error_in_ranges
    assert (len(g.zᵃᵃᶜ), len(g.zᵃᵃᶠ)) == (66, 67)


def test_xnode_ynode_znode_and_spacings_as_the_reference_tests_them(ocn_host):
    """test_grids.jl:184-241 on the regularly spaced grid and on the grid whose z is given as the same faces (x, y stay regular: the
    accelerated path takes stretched z only)"""
    ocn = ocn_host
    N, pi = 3, np.pi
    topo = (ocn.Periodic, ocn.Periodic, ocn.Bounded)
    domain = np.linspace(0, pi, N + 1)
    for z in ((0, pi), domain):
        g = ocn.RectilinearGrid(None, size=(N, N, N), x=(0, pi), y=(0, pi), z=z, topology=topo)
        xc, yc, zc = (a.ravel() for a in g.nodes((ocn.Center,) * 3))
        xf, yf, zf = (a.ravel() for a in g.nodes((ocn.Face,) * 3))
        assert np.allclose([xc[1], yc[1], zc[1]], pi / 2, rtol=1e-15) and np.allclose([xf[1], yf[1], zf[1]], pi / 3, rtol=1e-15)
        dz = np.asarray(g.Δzᵃᵃᶜ[g.Hz:g.Hz + N])
        assert np.allclose([g.Δxᶜᵃᵃ, g.Δyᵃᶜᵃ, dz.min()], pi / 3, rtol=1e-15) and np.allclose(dz, pi / N, rtol=1e-15)
        assert np.allclose(np.asarray(g.Δzᵃᵃᶠ[g.Hz:g.Hz + N + 1]), pi / N, rtol=1e-15) and g.Δxᶠᵃᵃ == g.Δxᶜᵃᵃ and g.Δyᵃᶠᵃ == g.Δyᵃᶜᵃ


@pytest.mark.parametrize("N", [16, 17])
def test_rectilinear_grid_correct_spacings_of_a_stretched_z(ocn_host, N):
    """test_grids.jl:440-487 (the z part: x = collect(0:N) is regular, the quadratic y is outside the accelerated path): tanh-like faces,
    S = 3"""
    ocn = ocn_host
    S = 3

    def zf(k):
        return np.tanh(S * (2 * (k - 1) / N - 1)) / np.tanh(S)
    g = ocn.RectilinearGrid(None, size=(N, N, N), x=(0, N), y=(0, N), z=zf, topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    H = g.Hz
    k = np.arange(1, N + 2)
    assert g.Δxᶜᵃᵃ == 1 and g.Δxᶠᵃᵃ == 1
    zc = (zf(k[:-1]) + zf(k[1:])) / 2
    assert np.allclose(g.zᵃᵃᶠ[H:H + N + 1], zf(k)) and np.allclose(g.zᵃᵃᶜ[H:H + N], zc)
    assert np.allclose(g.Δzᵃᵃᶜ[H:H + N], zf(k[1:]) - zf(k[:-1]))
    assert np.allclose(g.Δzᵃᵃᶠ[H + 1:H + N], zc[1:] - zc[:-1])          # Δzᵃᵃᶠ[2:N]; [1] involves a halo point


def test_grid_lengths_areas_and_volumes_of_test_operators(ocn_host, oracle):
    """test/test_operators.jl:158-224: size (1, 1, 1), extent (π, 2π, 3π): every Δx == π, Δy == 2π, Δz == 3π, Ax == 6π², Ay == 3π²,
    Az == 2π², V == 6π³ (exact equalities in the reference) -- with the products formed as the metric tables of the kernels form them
    (Ax = Δy Δz, Ay = Δx Δz, Az = Δx Δy, V = Az Δz: spacings_and_areas_and_volumes.jl:308-378; csrc/ocn_api.hip grid tables)"""
    ocn = ocn_host
    pi = np.pi
    g = ocn.RectilinearGrid(None, size=(1, 1, 1), extent=(pi, 2 * pi, 3 * pi), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    go = oracle.Grid((1, 1, 1), topology=(0, 0, 1), x=(0.0, pi), y=(0.0, 2 * pi), z=(-3 * pi, 0.0), halo=(1, 1, 1))
    for dx, dy, dz in ((g.Δxᶜᵃᵃ, g.Δyᵃᶜᵃ, float(g.Δzᵃᵃᶜ[g.Hz])), (go.dc[0][1], go.dc[1][1], go.dc[2][1])):
        assert (dx, dy, dz) == (pi, 2 * pi, 3 * pi)
        assert dy * dz == 6 * pi ** 2 and dx * dz == 3 * pi ** 2 and dx * dy == 2 * pi ** 2 and (dx * dy) * dz == 6 * pi ** 3


def test_grid_reconstruction_from_constructor_arguments(ocn_host):
    """test/test_grid_reconstruction.jl:12-45 (regular RectilinearGrid) and :47-95 with the stretched direction this library takes (z as a
    function): the grid rebuilt from constructor_arguments(grid) -- what with_halo / on_architecture use (rectilinear_grid.jl:404-449) --
    has the same size, halo, spacings and face / centre coordinates"""
    ocn = ocn_host
    P, B = ocn.Periodic, ocn.Bounded
    pi = np.pi
    originals = [ocn.RectilinearGrid(None, size=(4, 6, 8), extent=(2 * pi, 3 * pi, 4 * pi), topology=(P, B, B), halo=(2, 3, 2)),
                 ocn.RectilinearGrid(None, size=(8, 4, 8), x=(0, 1), y=(0, 1), z=lambda k: -1.0 + (k - 1) / 8, topology=(B, B, B), halo=(1, 1, 1))]
    for g in originals:
        r = ocn.with_halo(g.halo_size, g)
        assert r is not g and r.size == g.size and r.halo_size == g.halo_size and r.topology == g.topology
        assert (r.Δxᶠᵃᵃ, r.Δyᵃᶠᵃ) == (g.Δxᶠᵃᵃ, g.Δyᵃᶠᵃ) and np.array_equal(r.Δzᵃᵃᶠ, g.Δzᵃᵃᶠ) and np.array_equal(r.Δzᵃᵃᶜ, g.Δzᵃᵃᶜ)
        import unicodedata
        for name in ("xᶠᵃᵃ", "xᶜᵃᵃ", "yᵃᶠᵃ", "yᵃᶜᵃ", "zᵃᵃᶠ", "zᵃᵃᶜ"):
            key = unicodedata.normalize("NFKC", name)             # Python normalises identifiers: the attribute written .xᶠᵃᵃ is stored under this key
            assert np.array_equal(getattr(r, key), getattr(g, key)), name
        bigger = ocn.with_halo((4, 4, 4), g)                      # with_halo: the same interior nodes under another halo, to the round-off of
        # a range that starts elsewhere (the node coordinates are Julia ranges over F- .. F+, which move with the halo)
        assert bigger.halo_size == (4, 4, 4)
        for loc in ((ocn.Center,) * 3, (ocn.Face,) * 3):
            for a, b in zip(bigger.nodes(loc), g.nodes(loc)):
                assert np.allclose(a, b, rtol=4 * np.finfo(float).eps, atol=4 * np.finfo(float).eps)
