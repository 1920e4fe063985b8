"""The C ABI never aborts ("C ABI returns int status ... never abort"): every entry point of include/ocn_mi355x.h that takes
a pointer is called with all-NULL / all-zero arguments on the GPU box, in a child process (a crash would take the test runner down);
each call must return -- with a non-zero status where an object or an array is required -- and leave a message in ocn_last_error()."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
from oldoceananigans_jl_amd import _lib
L = _lib.lib()
assert L.ocn_init(0) == 0
out = {}
for name, (restype, argtypes) in sorted(_lib.SYMBOLS.items()):
    if restype is not C.c_int or not any(hasattr(a, "contents") or a is C.c_void_p or a is C.c_char_p for a in argtypes):
        continue
    if name in ("ocn_init",):
        continue
    args = []
    for a in argtypes:
        if a in (C.c_int, C.c_size_t, C.c_long, C.c_int64):
            args.append(0)
        elif a is C.c_double:
            args.append(0.0)
        else:
            args.append(None)
    rc = getattr(L, name)(*args)
    msg = L.ocn_last_error()
    out[name] = [int(rc), (msg or b"").decode(errors="replace")[:80]]
    print(name, rc, flush=True)
print("RESULT " + json.dumps(out))
"""


def test_every_entry_point_returns_on_null_arguments():
    env = dict(os.environ, OCN_TEST_NO_TORCH="1")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-600:], r.stderr[-1200:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert len(out) > 80
    # NULL is a valid argument only for releasing / waiting on nothing, for process-wide settings and for a copy of zero bytes
    may_succeed = ("destroy", "free", "close", "set_option", "set_stream", "own_stream", "wait", "debug", "memcpy", "memset")
    bad = {n: v for n, v in out.items() if v[0] == 0 and not any(t in n for t in may_succeed)}
    assert not bad, bad
    assert all(v[1] for n, v in out.items() if v[0] != 0), "a failing call left no message"


def test_entry_points_the_host_mirror_does_not_use(ocn, oracle, arch):
    """the raw C-ABI entry points no host-mirror call goes through, called directly: ocn_version / ocn_stream / ocn_memset_zero /
    ocn_grid_parent_size / ocn_poisson_kind, ocn_compute_source_term (complex storage of the reference, against the oracle),
    ocn_ab2_step (against ab2_step_field!'s formula and the oracle), ocn_dist_set_layout + ocn_dist_info (rank = ix * Ry + iy,
    periodic wrap of the four neighbours, distributed_architectures.jl:354-434)"""
    import ctypes as C

    import numpy as np
    from oldoceananigans_jl_amd import _lib
    L = _lib.lib()
    assert L.ocn_version().decode().count(".") >= 1 and L.ocn_stream()
    grid = ocn.RectilinearGrid(arch, size=(6, 5, 4), x=(0, 1), y=(0, 1), z=[0, 0.1, 0.3, 0.6, 1], topology=(ocn.Periodic, ocn.Bounded, ocn.Bounded))
    P = (C.c_int * 3)()
    for loc, want in (((0, 0, 0), (12, 11, 10)), ((1, 0, 0), (12, 11, 10)), ((0, 1, 0), (12, 12, 10)), ((0, 0, 1), (12, 11, 11))):
        _lib.check(L.ocn_grid_parent_size(grid.handle, (C.c_int * 3)(*loc), P))
        assert tuple(P) == want
    c = ocn.CenterField(grid)
    c.set(3.0)
    _lib.check(L.ocn_memset_zero(c.data, c.nbytes))
    assert not c.parent().any()
    assert L.ocn_poisson_kind(ocn.FourierTridiagonalPoissonSolver(grid).handle) == 1
    g2 = ocn.RectilinearGrid(arch, size=(6, 5, 4), extent=(1, 1, 1), topology=(ocn.Periodic, ocn.Bounded, ocn.Bounded))
    assert L.ocn_poisson_kind(ocn.FFTBasedPoissonSolver(g2).handle) == 0
    # source term in the reference's complex storage
    rng = np.random.default_rng(0)
    go = oracle.Grid((6, 5, 4), topology=(0, 1, 1), x=(0.0, 1.0), y=(0.0, 1.0), z=np.array([0, 0.1, 0.3, 0.6, 1.0]))
    U, A = [], []
    for make, loc in ((ocn.XFaceField, "u"), (ocn.YFaceField, "v"), (ocn.ZFaceField, "w")):
        f = make(grid)
        a = np.asfortranarray(rng.standard_normal(f.shape))
        f.set_parent(a)
        ocn.fill_halo_regions(f)
        go.fill_halo_regions(a, oracle.LOC[loc])
        U.append(f)
        A.append(a)
    rhs = C.c_void_p()
    _lib.check(L.ocn_malloc(C.byref(rhs), 6 * 5 * 4 * 16))
    for weighted in (0, 1):
        _lib.check(L.ocn_compute_source_term(grid.handle, U[0].data, U[1].data, U[2].data, rhs, weighted))
        out = np.empty((6, 5, 4), dtype=np.complex128, order="F")
        _lib.check(L.ocn_memcpy_d2h(out.ctypes.data, rhs, out.nbytes))
        want = go.source_term(*A, weight_by_dz=bool(weighted))
        assert np.array_equal(out.real, np.asarray(want).real) and not out.imag.any()
    _lib.check(L.ocn_free(rhs))
    # ab2_step_field! on (u, c): U += Δt ((3/2 + χ) Gⁿ - (1/2 + χ) G⁻), wall-normal u faces excluded? (x Periodic here: every cell)
    flds = [ocn.XFaceField(grid), ocn.CenterField(grid)]
    Gn, Gm = [ocn.XFaceField(grid), ocn.CenterField(grid)], [ocn.XFaceField(grid), ocn.CenterField(grid)]
    vals = []
    for group in (flds, Gn, Gm):
        vals.append([])
        for f in group:
            a = np.asfortranarray(rng.standard_normal(f.shape))
            f.set_parent(a)
            vals[-1].append(a)
    ptrs = lambda fs: (C.c_void_p * len(fs))(*[f.data for f in fs])               # noqa: E731
    locs = (C.c_int * 6)(1, 0, 0, 0, 0, 0)
    dt, chi = 0.37, 0.1
    _lib.check(L.ocn_ab2_step(grid.handle, ptrs(flds), ptrs(Gn), ptrs(Gm), locs, 2, dt, chi))
    for n, f in enumerate(flds):
        got = f.interior()
        H = 3
        core = (slice(H, H + got.shape[0]), slice(H, H + got.shape[1]), slice(H, H + got.shape[2]))
        u0, gn, gm = vals[0][n][core], vals[1][n][core], vals[2][n][core]
        want = u0 + dt * ((1.5 + chi) * gn - (0.5 + chi) * gm)
        assert np.allclose(got, want, rtol=1e-14, atol=1e-15), n
    # Partition(Rx, Ry) layout of a communicator: rank = ix * Ry + iy and the wrapped neighbours
    from loopback import PointerLoopbackWorld
    from oldoceananigans_jl_amd import distributed as dist
    world = PointerLoopbackWorld(6, L)
    ctx = dist.Distributed.transport(arch, world.collectives(4), 6, 4)             # rank 4 of 6 = (ix, iy) = (2, 0) for Partition(3, 2)
    _lib.check(L.ocn_dist_set_layout(ctx.handle, 3, 2))
    w, r, west, east = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _lib.check(L.ocn_dist_info(ctx.handle, C.byref(w), C.byref(r), C.byref(west), C.byref(east)))
    assert (w.value, r.value, west.value, east.value) == (6, 4, 2, 0)             # west = (1, 0) -> 2, east wraps to (0, 0) -> 0
    assert L.ocn_dist_set_layout(ctx.handle, 4, 2) != 0                           # 4 x 2 != 6 ranks
    ctx.close()
