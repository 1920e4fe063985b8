"""GPU: the partitioned time-step INSIDE the library (include/ocn_mi355x.h, distributed group: ocn_dist_create, ocn_dist_model_create,
ocn_model_time_step on a distributed model) -- the product's N > 1 path.

  * over RCCL itself with a communicator of one rank that is its own neighbour (ocn_dist_set_self_loop): the whole partitioned code
    path runs through ncclSend / ncclRecv / ncclAllGather on the one-GPU box, against the single-GPU model;
  * with R virtual ranks (threads sharing the card) over the caller-supplied transport (ocn_dist_create_transport, tests/loopback.py),
    against the single-GPU model: same-peer case R = 2, R = 4, odd local sizes, Bounded / stretched z (transposing
    Fourier-tridiagonal solver), the configs[4] physics.

Reference tests mirrored: test_distributed_models.jl:334-404 (rank ids in the halos), test_distributed_poisson_solvers.jl:70-163 and
the distributed-vs-serial agreement of test_distributed_models.jl."""
import ctypes as C
import threading

import numpy as np
import pytest

from dist_worker import analytic
from test_gpu_distributed import _bcs, _closure, _tracers_and_buoyancy, _z_and_topology

pytestmark = pytest.mark.gpu


def _own_stream():
    from oldoceananigans_jl_amd import _lib
    _lib.check(_lib.lib().ocn_own_stream())


def _xb(ocn, topo, xbounded, ybounded=False):
    return ((ocn.Bounded if xbounded else topo[0]), (ocn.Bounded if ybounded else topo[1]), topo[2])


def _single_gpu(ocn, arch, size, zkind, nsteps, xbounded=False, ybounded=False):
    z, topo = _z_and_topology(ocn, zkind, size[2])
    topo = _xb(ocn, topo, xbounded, ybounded)
    grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
    model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), boundary_conditions=_bcs(ocn, zkind), closure=_closure(ocn, zkind),
                                    buoyancy=_tracers_and_buoyancy(ocn, zkind)[1], coriolis=ocn.FPlane(f=0.5) if zkind == "stretched" else None)
    ocn.set_model(model, **{n: analytic(n, *grid.nodes(f.loc)) for n, f in model.fields().items()})
    dt = 0.1 * grid.Δxᶜᵃᵃ / 0.6
    for _ in range(nsteps):
        ocn.time_step(model, dt)
    out = {n: f.parent() for n, f in model.fields().items()}
    out["p"] = model.pressures.pNHS.parent()
    return out, model.clock.time, dt


def _library_model(ocn, dist, ctx, size, zkind, xbounded=False, ybounded=False, partition=None):
    z, topo = _z_and_topology(ocn, zkind, size[2])
    topo = _xb(ocn, topo, xbounded, ybounded)
    grid = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo, partition=partition)
    model = dist.LibraryDistributedModel(grid=grid, tracers=("T", "S"), boundary_conditions=_bcs(ocn, zkind), closure=_closure(ocn, zkind),
                                         buoyancy=_tracers_and_buoyancy(ocn, zkind)[1],
                                         coriolis=ocn.FPlane(f=0.5) if zkind == "stretched" else None)
    ocn.set_model(model, **{n: analytic(n, *grid.global_nodes(f.loc)) for n, f in model.fields().items()})
    return grid, model


def _compare(out, ref, r, nxl, size, offset=None, joffset=0):
    i0 = r * nxl if offset is None else offset
    for name, a in out.items():
        n = a.shape[0] - 6                  # nxl, or nxl + 1 for the Face-in-x field of a LeftConnected rank (its wall face)
        want = ref[name][3 + i0:3 + i0 + n, 3 + joffset:3 + joffset + a.shape[1] - 6, 3:3 + a.shape[2] - 6]
        scale = np.abs(ref[name]).max()
        err = np.abs(a[3:-3, 3:-3, 3:-3] - want).max() / scale
        assert err <= 1e-12, (r, name, err, int(np.isnan(a).sum()))


@pytest.mark.parametrize("size,zkind,substructured,async_halos", [
    ((32, 16, 8), "periodic", 1, 1), ((32, 16, 8), "periodic", 0, 1), ((32, 12, 10), "stretched", 1, 1), ((384, 8, 8), "periodic", 1, -1),
    ((32, 16, 8), "periodic", 1, 0)])
def test_library_self_loop_over_rccl_equals_single_gpu(ocn, arch, size, zkind, substructured, async_halos):
    """RCCL communicator of ONE rank that is its own west and east neighbour: pack / ncclSend + ncclRecv on the communication
    stream / unpack, the exchange started from make_pressure_correction!, thin exchanges, ncclAllGather of the interface values (or
    the two all-to-alls as grouped send / recv) -- the fields equal the single-GPU model's"""
    from oldoceananigans_jl_amd import _lib, distributed as dist
    _own_stream()
    ocn.set_option("dist_substructured", substructured)
    try:
        uid = C.create_string_buffer(128)
        _lib.check(_lib.lib().ocn_dist_unique_id(uid))
        ctx = dist.Distributed.rccl(arch, uid, 1, 0, self_loop=True)
        grid, model = _library_model(ocn, dist, ctx, size, zkind)
        assert grid.local.topology[0] is ocn.FullyConnected
        model.set_option("async_halos", async_halos)
        ref, time, dt = _single_gpu(ocn, arch, size, zkind, 3)
        for _ in range(3):
            ocn.time_step(model, dt)
        assert model.clock.time == time and model.clock.iteration == 3
        assert ocn.max_abs_divergence(model) < 5e-8
        out = {n: f.parent() for n, f in model.fields().items()}
        out["p"] = model.pressures.pNHS.parent()
        model.close()
        ctx.close()
    finally:
        ocn.set_option("dist_substructured", 1)
    _compare(out, ref, 0, size[0], size)
    for name in ("u", "v", "w", "T", "S"):      # the x halos hold bit-exact copies of the other side's interior columns
        assert np.array_equal(out[name][:3, 3:-3, 3:-3], out[name][-6:-3, 3:-3, 3:-3]), name


@pytest.mark.parametrize("size,R", [((32, 16, 8), 1), ((64, 16, 16), 2), ((384, 8, 8), 2), ((1536, 8, 8), 2), ((64, 16, 8), 1), ((256, 8, 16), 2)])
def test_library_x_solve_layouts_agree(ocn, arch, size, R):
    """the x-fastest substructured solve (paired z transform in LDS + Thomas scans over the lanes of a wave: one line per wave with 4 and 16
    elements per lane -- local Nx = 192, 768 -- or, for short lines, 8 / 4 / 2 lines per wave -- local Nx = 32, 64, 128) against the z-fastest one (option dist_xfast = 0) and, with and without the pressure step
    that skips the fills / copies between its stages (fused_step), against the single-GPU model: three RK3 steps, 1e-12"""
    from oldoceananigans_jl_amd import _lib, distributed as dist
    _own_stream()
    ref, time, dt = _single_gpu(ocn, arch, size, "periodic", 3)
    nxl = size[0] // R
    for xfast, fused in ((1, 1), (1, 0), (0, 1), (0, 0)):
        ocn.set_option("dist_xfast", xfast)
        try:
            if R == 1:
                uid = C.create_string_buffer(128)
                _lib.check(_lib.lib().ocn_dist_unique_id(uid))
                ctx = dist.Distributed.rccl(arch, uid, 1, 0, self_loop=True)
                grid, model = _library_model(ocn, dist, ctx, size, "periodic")
                assert model.get_option("dist_poisson_layout") in ((4,) if xfast else (1, 2, 3))
                model.set_option("fused_step", fused)
                assert model.get_option("fused_step") == fused
                for _ in range(3):
                    ocn.time_step(model, dt)
                results = [({n: f.parent() for n, f in model.fields().items()} | {"p": model.pressures.pNHS.parent()}, ocn.max_abs_divergence(model),
                            model.clock.time, None)]
                model.close()
                ctx.close()
            else:
                results = _run_library_ranks(ocn, arch, R, size, 3, "periodic", {"fused_step": fused})
        finally:
            ocn.set_option("dist_xfast", 1)
        for r, (out, div, t, _off) in enumerate(results):
            assert div < 5e-8 and t == time, (xfast, fused)
            _compare(out, ref, r, nxl, size)


def test_seeded_random_x_slab_partitions_match_single_gpu(ocn, arch):
    """eight seeded random x-slab runs through the in-library partitioned step (threads sharing the card) against the single-GPU model:
    2 .. 4 ranks, local widths 7 .. 40 (regular: every solver path -- x-fastest / z-fastest substructured, transposing), Ny (a multiple
    of the rank count, distributed_fft_based_poisson_solver.jl:218-226) and Nz at random, one of the four physics presets (triply periodic; Bounded z with ScalarDiffusivity + buoyancy;
    stretched z with Coriolis and Flux / Value / Gradient conditions; the configs[4] physics), three RK3 steps, 1e-12"""
    _own_stream()
    rng = np.random.default_rng(31337)
    for case in range(8):
        R = int(rng.integers(2, 5))
        nxl = int(rng.choice([8, 16, 32, 64, 7, 12, 20, 40]))
        ny = int(rng.choice([8, 12, 16, 24])) if rng.random() < 0.7 else int(rng.integers(6, 20))
        ny += (-ny) % R                                   # Ny divisible by the number of ranks, as the reference's transposing solvers require
        nz = int(rng.choice([8, 16])) if rng.random() < 0.5 else int(rng.integers(6, 14))
        zkind = str(rng.choice(["periodic", "bounded", "stretched", "amd"]))
        size = (R * nxl, ny, nz)
        ref, time, dt = _single_gpu(ocn, arch, size, zkind, 3)
        results = _run_library_ranks(ocn, arch, R, size, 3, zkind, {})
        for r, (out, div, t, _off) in enumerate(results):
            assert div < 5e-8 and t == time, (case, R, size, zkind)
            try:
                _compare(out, ref, r, nxl, size)
            except AssertionError as e:
                raise AssertionError((case, R, size, zkind, str(e)))


def test_fused_source_term_and_z_transform_is_bit_identical(ocn, arch):
    """x-fastest substructured solve: the source term written straight into the LDS line buffer of the paired z transform
    (source_paired_zline_r2c_kernel) against the two separate kernels (option dist_fuse_source = 0): same expressions on the same operands,
    so every field and the pressure after three steps bit for bit (one rank, self-loop over RCCL; 256 and 512-point z lines: both line widths)"""
    from oldoceananigans_jl_amd import _lib, distributed as dist
    _own_stream()
    for size in ((64, 32, 256), (32, 16, 512)):
        outs = []
        for fuse in (1, 0):
            ocn.set_option("dist_fuse_source", fuse)
            try:
                uid = C.create_string_buffer(128)
                _lib.check(_lib.lib().ocn_dist_unique_id(uid))
                ctx = dist.Distributed.rccl(arch, uid, 1, 0, self_loop=True)
                grid, model = _library_model(ocn, dist, ctx, size, "periodic")
                assert model.get_option("dist_poisson_layout") == 4 and model.get_option("fused_step") == 1
                for _ in range(3):
                    ocn.time_step(model, 1e-3)
                outs.append({n: f.parent() for n, f in model.fields().items()} | {"p": model.pressures.pNHS.parent()})
                assert ocn.max_abs_divergence(model) < 5e-8
                model.close()
                ctx.close()
            finally:
                ocn.set_option("dist_fuse_source", 1)
        for n in outs[0]:
            assert np.array_equal(outs[0][n], outs[1][n]), (size, n)


def test_library_collectives_over_rccl_world_1(ocn, arch):
    """the raw collectives of the boundary with a one-rank RCCL communicator: an exchange swaps the sides (what leaves through the
    west side arrives in the east halo), all-to-all / all-gather are copies, the reduction returns its argument"""
    from oldoceananigans_jl_amd import _lib
    L = _lib.lib()
    _own_stream()
    uid = C.create_string_buffer(128)
    _lib.check(L.ocn_dist_unique_id(uid))
    d = C.c_void_p()
    _lib.check(L.ocn_dist_create(C.byref(d), uid, 1, 0))
    w, r, we, ea = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _lib.check(L.ocn_dist_info(d, C.byref(w), C.byref(r), C.byref(we), C.byref(ea)))
    assert (w.value, r.value, we.value, ea.value) == (1, 0, 0, 0)
    n = 1000
    bufs = []
    for _ in range(4):
        p = C.c_void_p()
        _lib.check(L.ocn_malloc(C.byref(p), 8 * n))
        bufs.append(p)
    ws, es, wr, er = bufs
    a, b = np.arange(n, dtype=np.float64), -np.arange(n, dtype=np.float64)
    _lib.check(L.ocn_memcpy_h2d(ws, a.ctypes.data, 8 * n))
    _lib.check(L.ocn_memcpy_h2d(es, b.ctypes.data, 8 * n))
    _lib.check(L.ocn_dist_exchange_start(d, ws, es, wr, er, n))
    _lib.check(L.ocn_dist_exchange_wait(d))
    got_w, got_e = np.empty(n), np.empty(n)
    _lib.check(L.ocn_memcpy_d2h(got_w.ctypes.data, wr, 8 * n))
    _lib.check(L.ocn_memcpy_d2h(got_e.ctypes.data, er, 8 * n))
    assert np.array_equal(got_e, a) and np.array_equal(got_w, b)
    _lib.check(L.ocn_dist_all_to_all(d, ws, wr, n))
    _lib.check(L.ocn_dist_all_gather(d, es, er, n))
    _lib.check(L.ocn_memcpy_d2h(got_w.ctypes.data, wr, 8 * n))
    _lib.check(L.ocn_memcpy_d2h(got_e.ctypes.data, er, 8 * n))
    assert np.array_equal(got_w, a) and np.array_equal(got_e, b)
    v = C.c_double(-3.25)
    _lib.check(L.ocn_dist_allreduce_max(d, C.byref(v)))
    assert v.value == -3.25
    _lib.check(L.ocn_dist_barrier(d))
    for p in bufs:
        _lib.check(L.ocn_free(p))
    _lib.check(L.ocn_dist_destroy(d))


def _run_library_ranks(ocn, arch, R, size, nsteps, zkind, options, xbounded=False, ybounded=False, partition=None, probe=None):
    from oldoceananigans_jl_amd import _lib, distributed as dist
    from loopback import PointerLoopbackWorld
    world = PointerLoopbackWorld(R, _lib.lib())
    results, errors = [None] * R, []
    dt = 0.1 * (2.0 / size[0]) / 0.6

    def worker(rank):
        try:
            ctx = dist.Distributed.transport(arch, world.collectives(rank), R, rank)
            grid, model = _library_model(ocn, dist, ctx, size, zkind, xbounded, ybounded, partition)
            for k, v in options.items():
                model.set_option(k, v)
            if probe is not None:
                probe(model)
            for _ in range(nsteps):
                ocn.time_step(model, dt)
            div = ocn.max_abs_divergence(model)
            out = {n: f.parent() for n, f in model.fields().items()}
            out["p"] = model.pressures.pNHS.parent()
            # test/test_distributed_models.jl:334-404: fields filled with the rank id; after update_state! the x halos hold the neighbours' ids
            for n, f in enumerate(model.fields().values()):
                f.set_parent(np.full(f.shape, 100.0 * n + rank))
            ocn.update_state(model, False)
            Rx_, Ry_ = partition if partition else (R, 1)
            ix_, iy_ = rank // Ry_, rank % Ry_
            west, east = ((ix_ - 1) % Rx_) * Ry_ + iy_, ((ix_ + 1) % Rx_) * Ry_ + iy_
            south, north = ix_ * Ry_ + (iy_ - 1) % Ry_, ix_ * Ry_ + (iy_ + 1) % Ry_
            # a wall side has no neighbour: the halo keeps what the local boundary fill leaves there (built from the rank's own id)
            wall_w, wall_e = xbounded and ix_ == 0, xbounded and ix_ == Rx_ - 1
            wall_s, wall_n = ybounded and Ry_ > 1 and iy_ == 0, ybounded and Ry_ > 1 and iy_ == Ry_ - 1
            for n, f in enumerate(model.fields().values()):
                a = f.parent()
                if not wall_w:
                    assert np.all(a[:3, 3:-3, 3:-3] == 100 * n + west), (rank, n)
                if not wall_e:
                    assert np.all(a[-3:, 3:-3, 3:-3] == 100 * n + east), (rank, n)
                if Ry_ > 1:         # pencils: y halos from the south / north ranks, corners from the diagonal ones (two hops)
                    if not wall_s:
                        assert np.all(a[3:-3, :3, 3:-3] == 100 * n + south), (rank, n)
                    if not wall_n:
                        assert np.all(a[3:-3, -3:, 3:-3] == 100 * n + north), (rank, n)
                    if Rx_ > 1:
                        sw = ((ix_ - 1) % Rx_) * Ry_ + (iy_ - 1) % Ry_
                        ne = ((ix_ + 1) % Rx_) * Ry_ + (iy_ + 1) % Ry_
                        if not (wall_w or wall_s):
                            assert np.all(a[:3, :3, 3:-3] == 100 * n + sw), (rank, n)
                        if not (wall_e or wall_n):
                            assert np.all(a[-3:, -3:, 3:-3] == 100 * n + ne), (rank, n)
                        se = ((ix_ + 1) % Rx_) * Ry_ + (iy_ - 1) % Ry_
                        nw = ((ix_ - 1) % Rx_) * Ry_ + (iy_ + 1) % Ry_
                        if not (wall_e or wall_s):
                            assert np.all(a[-3:, :3, 3:-3] == 100 * n + se), (rank, n)
                        if not (wall_w or wall_n):
                            assert np.all(a[:3, -3:, 3:-3] == 100 * n + nw), (rank, n)
            results[rank] = (out, div, model.clock.time, (grid.i_offset, grid.j_offset))
            model.close()
            ctx.close()
        except BaseException as e:          # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            world.barrier_obj.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    return results


@pytest.mark.parametrize("R,size,zkind,options", [
    (2, (32, 16, 8), "periodic", {}),                                   # both neighbours are the same rank
    (2, (32, 16, 8), "periodic", {"async_halos": 0}),
    (2, (32, 16, 8), "periodic", {"early_exchange": 0, "async_halos": 1}),   # interior / buffer split with Hx-wide strips
    (2, (32, 16, 8), "periodic", {"thin_halos": 0}),
    (4, (36, 12, 10), "periodic", {}),                                  # odd local Nx (9)
    (8, (64, 8, 8), "periodic", {}),                                    # the rank count of one MI355X node
    (3, (24, 9, 6), "periodic", {}),
    (2, (384, 8, 8), "periodic", {"early_exchange": 0, "async_halos": 1}),   # local Nx = 192: buffer strips one 64-lane tile wide
    (2, (32, 16, 8), "bounded", {}),                                    # distributed Fourier-tridiagonal solver + ScalarDiffusivity + buoyancy
    (8, (64, 8, 8), "stretched", {}),                                   # transposing solver over 8 ranks, Coriolis, boundary conditions
    (4, (32, 12, 10), "amd", {}),                                       # the configs[4] physics
])
def test_library_virtual_ranks_match_single_gpu(ocn, arch, R, size, zkind, options):
    _own_stream()
    nsteps = 3
    results = _run_library_ranks(ocn, arch, R, size, nsteps, zkind, options)
    ref, time, _ = _single_gpu(ocn, arch, size, zkind, nsteps)
    nxl = size[0] // R
    for r, (out, div, t, _off) in enumerate(results):
        assert div < 5e-8 and t == time
        _compare(out, ref, r, nxl, size)


@pytest.mark.parametrize("R,size,zkind", [
    (3, (25, 8, 6), "periodic"),          # 8, 8, 9 columns; Ny not divisible by R either
    (2, (13, 8, 6), "bounded"),           # 6, 7 columns: Fourier-tridiagonal solver + ScalarDiffusivity + buoyancy
    (4, (30, 8, 8), "stretched"),         # 7, 7, 7, 9 columns, stretched z, Coriolis, boundary conditions
    (4, (31, 8, 8), "amd"),               # the configs[4] physics on 7, 7, 7, 10 columns
])
def test_library_irregular_partition_matches_single_gpu(ocn, arch, R, size, zkind):
    """Nx not divisible by the number of ranks: local_size puts the remainder on the last rank (distributed_grids.jl:44-58,
    partition_coordinate partition_assemble.jl:63-76). The halo exchange is unchanged (Hx columns per side whatever the slab width); the
    pressure solve takes the gathered form (every rank assembles the global source term and runs the single-GPU solver)."""
    from oldoceananigans_jl_amd import distributed as dist
    assert dist.local_sizes(25, 3) == [8, 8, 9] and dist.local_sizes(24, 3) == [8, 8, 8] and dist.local_sizes(7, 4) == [1, 1, 1, 4]
    edges = [dist.partition_coordinate((0.0, 2.0), dist.local_sizes(25, 3), 3, r) for r in range(3)]
    assert edges[0][0] == 0.0 and all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:])) and abs(edges[-1][1] - 2.0) < 1e-15
    _own_stream()
    nsteps = 3
    results = _run_library_ranks(ocn, arch, R, size, nsteps, zkind, {})
    ref, time, _ = _single_gpu(ocn, arch, size, zkind, nsteps)
    sizes = dist.local_sizes(size[0], R)
    for r, (out, div, t, _off) in enumerate(results):
        assert div < 5e-8 and t == time
        _compare(out, ref, r, sizes[r], size, offset=sum(sizes[:r]))


@pytest.mark.parametrize("R,size,zkind,ybounded", [
    (3, (25, 8, 6), "periodic", False),     # (Bounded, Periodic, Periodic), 8 + 8 + 9 columns: Right-, Fully-, LeftConnected ranks
    (2, (16, 8, 6), "periodic", False),     # two ranks: a RightConnected and a LeftConnected one, no FullyConnected rank
    (4, (32, 8, 8), "bounded", False),      # (Bounded, Periodic, Bounded): Fourier-tridiagonal solver, ScalarDiffusivity, buoyancy
    (3, (24, 9, 6), "bounded", True),       # (Bounded, Bounded, Bounded): the one Bounded-x topology the reference's distributed solvers take
    (4, (30, 8, 8), "stretched", False),    # stretched z, Coriolis next to the walls, boundary conditions, irregular slabs
    (3, (24, 8, 8), "amd", False),          # the configs[4] physics next to walls
])
def test_library_bounded_partition_matches_single_gpu(ocn, arch, R, size, zkind, ybounded):
    """a Bounded partitioned direction: insert_connected_topology (distributed_grids.jl:339-346) gives the first rank a RightConnected
    local grid (wall on its west side), the last one a LeftConnected one (wall on the east side, Nx + 1 x-faces), the others
    FullyConnected; the advection scheme falls back next to the wall side only (topologically_conditional_interpolation.jl:54-70),
    boundary conditions fill the wall side, the ring has no wrap-around neighbour. Against the single-GPU model on the global grid."""
    _own_stream()
    from oldoceananigans_jl_amd import distributed as dist
    nsteps = 3
    results = _run_library_ranks(ocn, arch, R, size, nsteps, zkind, {}, xbounded=True, ybounded=ybounded)
    ref, time, _ = _single_gpu(ocn, arch, size, zkind, nsteps, xbounded=True, ybounded=ybounded)
    sizes = dist.local_sizes(size[0], R)
    for r, (out, div, t, _off) in enumerate(results):
        assert div < 5e-8 and t == time
        _compare(out, ref, r, sizes[r], size, offset=sum(sizes[:r]))
        assert out["u"].shape[0] == sizes[r] + 6 + (1 if r == R - 1 else 0)


@pytest.mark.parametrize("partition,size,zkind,xbounded", [
    ((2, 2), (16, 16, 8), "periodic", False),     # four pencils, every rank has four distinct neighbours + diagonals
    ((1, 3), (16, 18, 6), "periodic", False),     # y-slabs only: x stays Periodic locally
    ((3, 2), (25, 14, 6), "periodic", False),     # six ranks, irregular in x (8 + 8 + 9), 7 + 7 rows
    ((2, 2), (16, 13, 8), "bounded", False),      # z Bounded: Fourier-tridiagonal solver + diffusivity + buoyancy; 6 + 7 rows
    ((2, 2), (16, 12, 8), "amd", False),          # the configs[4] physics on pencils (eddy diffusivities extended into x AND y halos)
    ((2, 2), (16, 12, 8), "stretched", True),     # Bounded x + pencils: Right / LeftConnected columns of ranks
])
@pytest.mark.parametrize("ybounded", [False, True])
def test_library_pencil_partition_matches_single_gpu(ocn, arch, partition, size, zkind, xbounded, ybounded):
    """Partition(Rx, Ry) (row (f).4 of SURVEY.md 8: pencil decomposition + corner exchange): rank = ix * Ry + iy
    (distributed_architectures.jl:354-434), local grids connected in x and FullyConnected in y; a fill makes two hops (x, then y over
    the whole x extent) so the corners hold the diagonal neighbours' data like after fill_corners! (halo_communication.jl:137-162) --
    asserted with rank ids --; the pressure solve is the gathered one. Against the single-GPU model on the global grid.
    ybounded: the partitioned y direction is Bounded -- Right / Fully / LeftConnected rows of ranks (insert_connected_topology,
    distributed_grids.jl:339-346), walls, wall fallbacks and boundary conditions on the first and the last row only, Ny + 1 y-faces on
    the last row -- the (1, 4, 1) and (2, 2, 1) partitions of Bounded topologies in test_distributed_poisson_solvers.jl:123-148."""
    _own_stream()
    nsteps = 3
    R = partition[0] * partition[1]
    results = _run_library_ranks(ocn, arch, R, size, nsteps, zkind, {}, xbounded=xbounded, ybounded=ybounded, partition=partition)
    ref, time, _ = _single_gpu(ocn, arch, size, zkind, nsteps, xbounded=xbounded, ybounded=ybounded)
    for r, (out, div, t, (i0, j0)) in enumerate(results):
        assert div < 5e-8 and t == time
        _compare(out, ref, r, None, size, offset=i0, joffset=j0)


def test_seeded_random_pencil_and_wall_partitions_match_single_gpu(ocn, arch):
    """eight seeded random partitions with pencils and / or walls in the partitioned directions: Partition(Rx, Ry) with Rx, Ry in 1 .. 3 (at most
    six ranks share the card), x and y Periodic or Bounded at random, irregular local sizes (remainder on the last rank), one of the four
    physics presets -- the transposing pencil solver on triply periodic regular cases, the gathered solve elsewhere --, three RK3 steps, 1e-12"""
    _own_stream()
    rng = np.random.default_rng(2718)
    done = 0
    while done < 8:
        Rx, Ry = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        if Rx * Ry < 2 or Rx * Ry > 6:
            continue
        nx = Rx * int(rng.integers(7, 13)) + (int(rng.integers(0, 3)) if Rx > 1 else 0)      # the last rank takes the remainder
        ny = Ry * int(rng.integers(7, 11)) + (int(rng.integers(0, 3)) if Ry > 1 else 0)
        nz = int(rng.integers(6, 11))
        if Ry == 1:
            ny += (-ny) % Rx                              # x-slabs: the reference's transposing solvers need Ny divisible by the rank count
        zkind = str(rng.choice(["periodic", "bounded", "stretched", "amd"]))
        xb, yb = bool(rng.random() < 0.4), bool(rng.random() < 0.4 and Ry > 1)
        size = (nx, ny, nz)
        results = _run_library_ranks(ocn, arch, Rx * Ry, size, 3, zkind, {}, xbounded=xb, ybounded=yb, partition=(Rx, Ry))
        ref, time, _ = _single_gpu(ocn, arch, size, zkind, 3, xbounded=xb, ybounded=yb)
        for r, (out, div, t, (i0, j0)) in enumerate(results):
            assert div < 5e-8 and t == time, (done, (Rx, Ry), size, zkind, xb, yb)
            try:
                _compare(out, ref, r, None, size, offset=i0, joffset=j0)
            except AssertionError as e:
                raise AssertionError((done, (Rx, Ry), size, zkind, xb, yb, str(e)))
        done += 1


def test_library_transposing_solver_matches(ocn, arch):
    """the all-to-all form of the periodic solver (option dist_substructured = 0) through the library's orchestration"""
    _own_stream()
    ocn.set_option("dist_substructured", 0)
    try:
        results = _run_library_ranks(ocn, arch, 4, (36, 12, 10), 3, "periodic", {})
    finally:
        ocn.set_option("dist_substructured", 1)
    ref, time, _ = _single_gpu(ocn, arch, (36, 12, 10), "periodic", 3)
    for r, (out, div, t, _off) in enumerate(results):
        assert div < 5e-8 and t == time
        _compare(out, ref, r, 9, (36, 12, 10))


def _gpu_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("R,zkind,options", [(2, "periodic", ""), (2, "periodic", "early_exchange=0 async_halos=1"), (2, "periodic", "thin_halos=0"),
                                             (4, "periodic", ""), (2, "stretched", ""), (4, "stretched", "")])
def test_library_rccl_separate_processes(ocn, arch, tmp_path, R, zkind, options):
    """REAL ranks on REAL GPUs over RCCL: needs R cards, skipped on the one-GPU box (first executed on a multi-GPU lease). R = 2 is
    the same-peer case (both neighbours are one rank: two sends and two receives per pair in one group)"""
    if _gpu_count() < R:
        pytest.skip(f"needs {R} GPUs; this box has {_gpu_count()}")
    _separate_processes(ocn, arch, tmp_path, R, zkind, options, staged=False)


@pytest.mark.parametrize("R,zkind,options", [(2, "periodic", ""), (4, "periodic", ""), (2, "periodic", "dist_substructured=0"), (2, "stretched", ""),
                                             (2, "periodic", "early_exchange=0 async_halos=1")])
def test_library_separate_processes_on_one_card(ocn, arch, tmp_path, R, zkind, options):
    """the in-library partitioned step as REAL separate processes under torch.distributed.run -- rank-local coordinates and the
    communicator's rank / size from the environment -- sharing this box's one card, their collectives staged through the host over
    gloo (tests/host_staged.py through ocn_dist_create_transport). Fields after 3 RK3 steps against the single-GPU model."""
    _separate_processes(ocn, arch, tmp_path, R, zkind, options, staged=True)


def _separate_processes(ocn, arch, tmp_path, R, zkind, options, staged):
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    size = (64 * R, 16, 12) if not staged else (16 * R, 16, 12)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OCN_TEST_HOST_STAGED="1" if staged else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={R}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "gpu_lib_dist_worker.py"), str(tmp_path), "3", zkind] + [str(n) for n in size] + options.split()
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    z, topo = _z_and_topology(ocn, zkind, size[2])
    grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
    model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
    ocn.set_model(model, **{n: analytic(n, *grid.nodes(f.loc)) for n, f in model.fields().items()})
    dt = 0.1 * (2.0 / size[0]) / 0.6
    for _ in range(3):
        ocn.time_step(model, dt)
    ref = {n: f.parent() for n, f in model.fields().items()}
    ref["p"] = model.pressures.pNHS.parent()
    nxl = size[0] // R
    for r in range(R):
        d = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert float(d["div"]) < 5e-8 and float(d["time"]) == model.clock.time
        _compare({n: d[n] for n in ref}, ref, r, nxl, size)


# the reference's distributed Poisson test (test/test_distributed_poisson_solvers.jl:70-163) on its own partitions, sizes and topologies
_REF_FFT_CASES = [((44, 44, 8), (4, 1)), ((16, 44, 8), (4, 1)), ((44, 44, 8), (1, 4)), ((44, 16, 8), (1, 4)), ((16, 44, 8), (1, 4)),
                  ((22, 44, 8), (2, 2)), ((44, 22, 8), (2, 2))]
_REF_TRI_CASES = [((44, 44, 8), (4, 1)), ((16, 44, 8), (4, 1)), ((44, 44, 8), (1, 4)), ((16, 44, 8), (1, 4)), ((22, 8, 8), (2, 2)),
                  ((8, 22, 8), (2, 2))]


def _reference_distributed_poisson_case(ocn, arch, size, partition, topology, stretched_z):
    """divergence_free_poisson_solution / divergence_free_poisson_tridiagonal_solution (test_distributed_poisson_solvers.jl:70-116): random
    velocities on the partitioned grid, solve_for_pressure! with Δt = 1, ∇²ϕ ≈ R = ∇·U on every rank. Here through the partitioned
    MODEL's set! (projection with Δt = 1, set_nonhydrostatic_model.jl:52-57): u' = u - ∇ϕ, so ∇·u' = R - ∇²ϕ and the reference's
    assertion reads norm(∇·u') <= sqrt(eps) norm(R) on every rank."""
    from oldoceananigans_jl_amd import _lib, distributed as dist
    from loopback import PointerLoopbackWorld
    Rx, Ry = partition
    R = Rx * Ry
    L = 2 * np.pi
    topo = tuple(getattr(ocn, t) for t in topology)
    z = np.linspace(0.0, L, size[2] + 1) if stretched_z else (0.0, L)
    rng = np.random.default_rng(0)
    wall = [t == "Bounded" for t in topology]
    G = []
    for d in range(3):
        shape = list(size)
        shape[d] += 1 if wall[d] else 0
        a = rng.random(shape)
        if wall[d]:                                   # impenetrable walls (the default conditions' fill)
            idx = [slice(None)] * 3
            idx[d] = 0
            a[tuple(idx)] = 0.0
            idx[d] = -1
            a[tuple(idx)] = 0.0
        G.append(a)
    spacing = [L / n for n in size]
    Rg = np.zeros(size)
    for d in range(3):
        hi = np.roll(G[d], -1, axis=d) if not wall[d] else np.take(G[d], range(1, size[d] + 1), axis=d)
        lo = G[d] if not wall[d] else np.take(G[d], range(0, size[d]), axis=d)
        Rg += (hi - lo) / spacing[d]
    world = PointerLoopbackWorld(R, _lib.lib())
    errors, ok = [], [None] * R

    def worker(rank):
        try:
            ctx = dist.Distributed.transport(arch, world.collectives(rank), R, rank)
            grid = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, L), y=(0.0, L), z=z, topology=topo, partition=partition)
            model = dist.LibraryDistributedModel(grid=grid, tracers=("c",))
            i0, j0 = grid.i_offset, grid.j_offset
            vals = {}
            for name, g_ in zip("uvw", G):
                sx, sy, sz = model.fields()[name].interior().shape
                vals[name] = g_[i0:i0 + sx, j0:j0 + sy, :sz]
            ocn.set_model(model, **vals)
            nx, ny, nz = grid.local_size
            H = 3
            div = np.zeros((nx, ny, nz))
            for d, name in enumerate("uvw"):
                a = model.fields()[name].parent()
                inner = [slice(H, H + nx), slice(H, H + ny), slice(H, H + nz)]
                up = list(inner)
                up[d] = slice(H + 1, H + 1 + (nx, ny, nz)[d])
                div += (a[tuple(up)] - a[tuple(inner)]) / spacing[d]
            Rl = Rg[i0:i0 + nx, j0:j0 + ny, :]
            ok[rank] = (float(np.linalg.norm(div)), float(np.linalg.norm(Rl)))
            model.close()
            ctx.close()
        except BaseException as e:          # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            world.barrier_obj.abort()
    threads = [threading.Thread(target=worker, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for rank, (nd, nr) in enumerate(ok):
        assert nr > 0 and nd <= np.sqrt(np.finfo(float).eps) * nr, (size, partition, topology, rank, nd, nr)


@pytest.mark.parametrize("topology", [("Periodic", "Periodic", "Periodic"), ("Periodic", "Periodic", "Bounded"),
                                      ("Periodic", "Bounded", "Bounded"), ("Bounded", "Bounded", "Bounded")])
def test_reference_distributed_fft_poisson_solver_cases(ocn, arch, topology):
    """test_distributed_poisson_solvers.jl:118-136: (4, 1, 1), (1, 4, 1) and (2, 2, 1) ranks on the reference's sizes, four topologies.
    Not run: its two-dimensional cases (Nz = 1 in a non-Flat direction, refused by this library's model)."""
    _own_stream()
    for size, partition in _REF_FFT_CASES:
        _reference_distributed_poisson_case(ocn, arch, size, partition, topology, False)


def test_reference_distributed_fourier_tridiagonal_solver_cases(ocn, arch):
    """test_distributed_poisson_solvers.jl:138-152: (Bounded, Bounded, Bounded), z given as a face array (the Fourier-tridiagonal
    solver), the reference's partitions and sizes. Not run: (4, 44, 8) over (4, 1, 1) and (44, 4, 8) over (1, 4, 1) -- one column / row per
    rank is fewer than this library's halo of 3 (the reference runs them with halo 2)."""
    _own_stream()
    for size, partition in _REF_TRI_CASES:
        _reference_distributed_poisson_case(ocn, arch, size, partition, ("Bounded", "Bounded", "Bounded"), True)


def test_library_partition_with_conditions_on_the_diffusivity_fields(ocn, arch, monkeypatch):
    """the configs[4] physics with Value conditions on νₑ and κₑ.T at the bottom (boundary_conditions = (νₑ = ..., κₑ = (T = ...,))) on
    four x-slabs: the diffusivity fields are filled with their conditions on every rank (z sides) while their x halos are evaluated from
    the exchanged fields -- against the single-GPU model"""
    import test_gpu_dist_library as me
    base = me._bcs

    def with_kbcs(ocn_, zkind):
        b = dict(base(ocn_, zkind))
        F = ocn_.FieldBoundaryConditions
        b["νₑ"] = F(bottom=ocn_.ValueBoundaryCondition(1e-3))
        b["κₑ"] = {"T": F(bottom=ocn_.ValueBoundaryCondition(2e-3))}
        return b
    monkeypatch.setattr(me, "_bcs", with_kbcs)
    _own_stream()
    R, size, nsteps = 4, (32, 12, 10), 3
    results = _run_library_ranks(ocn, arch, R, size, nsteps, "amd", {})
    ref, time, _ = _single_gpu(ocn, arch, size, "amd", nsteps)
    for r, (out, div, t, _off) in enumerate(results):
        assert div < 5e-8 and t == time
        _compare(out, ref, r, size[0] // R, size)


def test_library_partition_amd_with_a_no_slip_bottom(ocn, arch, monkeypatch):
    """ValueBoundaryCondition(0) on u and v at the bottom with the AMD closure on four x-slabs: the viscous flux through the bottom face of
    a rank-edge column reads νₑ at (0, j, 0) / (Nx + 1, j, 0) -- cells of the x halo columns the rank evaluates itself AND of the z halo.
    A serial Periodic run copies the z-filled value there; the partitioned fill now writes it too (the reference's only_local_halos fill
    leaves it unwritten on a partitioned grid, halo_communication.jl:87-110 -- with a no-flux bottom the value multiplies zero, which is
    why the other AMD cases never saw it). Against the single-GPU model, 1e-12."""
    import test_gpu_dist_library as me
    base = me._bcs

    def no_slip(ocn_, zkind):
        b = dict(base(ocn_, zkind))
        F = ocn_.FieldBoundaryConditions
        b["u"] = F(top=ocn_.FluxBoundaryCondition(-1e-4), bottom=ocn_.ValueBoundaryCondition(0.0))
        b["v"] = F(bottom=ocn_.ValueBoundaryCondition(0.0))
        return b
    monkeypatch.setattr(me, "_bcs", no_slip)
    _own_stream()
    R, size, nsteps = 4, (32, 12, 10), 3
    results = _run_library_ranks(ocn, arch, R, size, nsteps, "amd", {})
    ref, time, _ = _single_gpu(ocn, arch, size, "amd", nsteps)
    for r, (out, div, t, _off) in enumerate(results):
        assert div < 5e-8 and t == time
        _compare(out, ref, r, size[0] // R, size)


def test_library_partition_with_a_flat_y_direction(ocn, arch):
    """(Periodic, Flat, Bounded) on two x-slabs: `size` / `extent` list the non-Flat directions only, like RectilinearGrid's; the library
    routes a Flat y to the gathered pressure solve. Against the single-GPU model."""
    from oldoceananigans_jl_amd import distributed as dist
    from loopback import PointerLoopbackWorld
    from oldoceananigans_jl_amd import _lib
    from helpers import tanh_faces
    _own_stream()
    R, nsteps = 2, 3
    topo = (ocn.Periodic, ocn.Flat, ocn.Bounded)
    size2, z = (32, 12), tanh_faces(12)
    gridS = ocn.RectilinearGrid(arch, size=size2, x=(0.0, 2.0), z=z, topology=topo)
    model = ocn.NonhydrostaticModel(grid=gridS, tracers=("T", "S"))
    ocn.set_model(model, **{n: analytic(n, *gridS.nodes(f.loc)) for n, f in model.fields().items()})
    dt = 0.1 * gridS.Δxᶜᵃᵃ / 0.6
    for _ in range(nsteps):
        ocn.time_step(model, dt)
    ref = {n: f.parent() for n, f in model.fields().items()} | {"p": model.pressures.pNHS.parent()}
    world = PointerLoopbackWorld(R, _lib.lib())
    results, errors = [None] * R, []

    def worker(rank):
        try:
            ctx = dist.Distributed.transport(arch, world.collectives(rank), R, rank)
            grid = dist.DistributedRectilinearGrid(ctx, size=size2, x=(0.0, 2.0), z=z, topology=topo)
            m = dist.LibraryDistributedModel(grid=grid, tracers=("T", "S"))
            assert m.get_option("dist_poisson_layout") == -2              # gathered solve
            ocn.set_model(m, **{n: analytic(n, *grid.global_nodes(f.loc)) for n, f in m.fields().items()})
            for _ in range(nsteps):
                ocn.time_step(m, dt)
            results[rank] = ({n: f.parent() for n, f in m.fields().items()} | {"p": m.pressures.pNHS.parent()}, ocn.max_abs_divergence(m))
            m.close()
            ctx.close()
        except BaseException as e:          # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            world.barrier_obj.abort()
    threads = [threading.Thread(target=worker, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    nxl = 32 // R
    for r, (out, div) in enumerate(results):
        assert div < 5e-8
        for name, a in out.items():
            hx = 3
            want = ref[name][hx + r * nxl:hx + (r + 1) * nxl]
            got = a[hx:hx + nxl]
            inner = (slice(None),) + tuple(slice(3, -3) if s > 1 else slice(None) for s in want.shape[1:])
            err = np.abs(got[inner] - want[inner]).max() / np.abs(ref[name]).max()
            assert err <= 1e-12, (r, name, err)


# test/test_distributed_transpose.jl:13-54 -- the reference's sizes and partitions (the topology plays no part in a transpose)
_TRANSPOSE_CASES = [((44, 44, 8), (4, 1)), ((16, 44, 8), (4, 1)), ((44, 44, 8), (1, 4)), ((44, 16, 8), (1, 4)), ((16, 44, 8), (1, 4)),
                    ((44, 16, 8), (2, 2)), ((16, 44, 8), (2, 2))]


@pytest.mark.parametrize("size,partition", _TRANSPOSE_CASES)
def test_reference_distributed_transpose_round_trip_is_bit_exact(ocn, arch, size, partition):
    """TransposableField + transpose_z_to_y! / y_to_x! / x_to_y! / y_to_z! (distributed_transpose.jl:25-95,185-191) on virtual ranks: the
    reference's test -- random ComplexF64 data makes the full cycle z -> y -> x -> y -> z and comes back bit for bit
    (test_distributed_transpose.jl:13-54) -- and, stronger, every intermediate configuration holds exactly its block of the global array:
    yfield = G[x block ix, :, z block iy], xfield = G[:, y block ix, z block iy] (the twin grids of transposable_field.jl:122-182)."""
    from oldoceananigans_jl_amd import _lib, distributed as dist
    from loopback import PointerLoopbackWorld
    _own_stream()
    L = _lib.lib()
    Rx, Ry = partition
    R = Rx * Ry
    Nx, Ny, Nz = size
    rng = np.random.default_rng(5)
    G = rng.random(size) + 1j * rng.random(size)                       # the global field, G[i, j, k]
    world = PointerLoopbackWorld(R, L)
    errors = []

    def read(ptr, shape):
        a = np.empty(shape, dtype=np.complex128, order="F")
        _lib.check(L.ocn_memcpy_d2h(a.ctypes.data, ptr, a.nbytes))
        return a

    def worker(rank):
        try:
            ctx = dist.Distributed.transport(arch, world.collectives(rank), R, rank)
            _lib.check(L.ocn_dist_set_layout(ctx.handle, Rx, Ry))
            tf = C.c_void_p()
            _lib.check(L.ocn_transposable_create(C.byref(tf), ctx.handle, Nx, Ny, Nz))
            zf, yf, xf = C.c_void_p(), C.c_void_p(), C.c_void_p()
            zs, ys, xs = (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)()
            _lib.check(L.ocn_transposable_fields(tf, C.byref(zf), C.byref(yf), C.byref(xf), zs, ys, xs))
            ix, iy = rank // Ry, rank % Ry
            nx, ny, nz, nyx = Nx // Rx, Ny // Ry, Nz // Ry, Ny // Rx
            assert tuple(zs) == (nx, ny, Nz) and tuple(ys) == (nx, Ny, nz) and tuple(xs) == (Nx, nyx, nz)
            mine = np.asfortranarray(G[ix * nx:(ix + 1) * nx, iy * ny:(iy + 1) * ny, :])
            _lib.check(L.ocn_memcpy_h2d(zf, mine.ctypes.data, mine.nbytes))
            _lib.check(L.ocn_transpose_z_to_y(tf))
            assert np.array_equal(read(yf, tuple(ys)), G[ix * nx:(ix + 1) * nx, :, iy * nz:(iy + 1) * nz]), "y-local block"
            _lib.check(L.ocn_transpose_y_to_x(tf))
            assert np.array_equal(read(xf, tuple(xs)), G[:, ix * nyx:(ix + 1) * nyx, iy * nz:(iy + 1) * nz]), "x-local block"
            _lib.check(L.ocn_transpose_x_to_y(tf))
            assert np.array_equal(read(yf, tuple(ys)), G[ix * nx:(ix + 1) * nx, :, iy * nz:(iy + 1) * nz]), "back in the y-local block"
            if Ry > 1:          # (on x-slabs yfield IS zfield: keep its content for the last comparison, scramble it otherwise)
                _lib.check(L.ocn_memset_zero(zf, mine.nbytes))
            _lib.check(L.ocn_transpose_y_to_z(tf))
            back = read(zf, tuple(zs))
            assert np.array_equal(back.real, mine.real) and np.array_equal(back.imag, mine.imag), "round trip"
            _lib.check(L.ocn_transposable_destroy(tf))
            ctx.close()
        except BaseException as e:          # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))
            world.barrier_obj.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("partition,size", [((2, 2), (16, 16, 8)), ((1, 4), (16, 16, 8)), ((4, 2), (32, 16, 8)), ((2, 4), (16, 32, 16))])
def test_library_pencil_models_take_the_transposing_solver(ocn, arch, partition, size):
    """pencil partitions of a triply Periodic regular grid run the reference's DistributedFFTBasedPoissonSolver -- z / y / x transforms on
    the TransposableField with its two transposes each way (distributed_fft_based_poisson_solver.jl:141-178) -- instead of the gathered
    solve (option dist_pencil_transposes = 0 brings that back): both against the single-GPU model, three RK3 steps, 1e-12"""
    _own_stream()
    R = partition[0] * partition[1]
    ref, time, _ = _single_gpu(ocn, arch, size, "periodic", 3)
    for transposes in (1, 0):
        ocn.set_option("dist_pencil_transposes", transposes)
        try:
            layouts = []
            results = _run_library_ranks(ocn, arch, R, size, 3, "periodic", {}, partition=partition, probe=lambda m: layouts.append(m.get_option("dist_poisson_layout")))
        finally:
            ocn.set_option("dist_pencil_transposes", 1)
        assert layouts == [-3 if transposes else -2] * R, layouts
        for r, (out, div, t, (i0, j0)) in enumerate(results):
            assert div < 5e-8 and t == time
            _compare(out, ref, r, None, size, offset=i0, joffset=j0)
