"""GPU: the product's distributed x-slab path (DeviceBackend: pack/unpack kernels, distributed FFT stages, async interior /
strip tendencies) with R virtual ranks on ONE GPU (tests/loopback.py), against the product's single-GPU model and the oracle.

Reference tests mirrored: halo exchange exactness (test_distributed_models.jl:334-404), transpose round trip
(test_distributed_transpose.jl:13-54), distributed Poisson residual (test_distributed_poisson_solvers.jl:70-163),
distributed-vs-serial model agreement."""
import threading

import numpy as np
import pytest

from dist_worker import analytic
from helpers import rel_err, tanh_faces

pytestmark = pytest.mark.gpu


def _z_and_topology(ocn, zkind, Nz):
    if zkind == "periodic":
        return (0.0, 1.0), (ocn.Periodic, ocn.Periodic, ocn.Periodic)
    if zkind == "bounded":
        return (-1.0, 0.0), (ocn.Periodic, ocn.Periodic, ocn.Bounded)
    return tanh_faces(Nz), (ocn.Periodic, ocn.Periodic, ocn.Bounded)


def _bcs(ocn, zkind):
    """config-5 style conditions on the Bounded z sides (constant Flux / Gradient / Value)"""
    F = ocn.FieldBoundaryConditions
    if zkind == "amd":      # the conditions of examples/ocean_wind_mixing_and_convection.jl, evaporation flux -rate S included
        return {"u": F(top=ocn.FluxBoundaryCondition(-1e-3)),
                "T": F(top=ocn.FluxBoundaryCondition(4e-3), bottom=ocn.GradientBoundaryCondition(0.01)),
                "S": F(top=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(b=-2.5e-3), field_dependencies="S"))}
    if zkind != "stretched":
        return None
    return {"u": F(top=ocn.FluxBoundaryCondition(-2e-3), bottom=ocn.ValueBoundaryCondition(0.0)),
            "T": F(top=ocn.FluxBoundaryCondition(5e-3), bottom=ocn.GradientBoundaryCondition(0.4))}


def _tracers_and_buoyancy(ocn, zkind):
    if zkind == "amd":
        return ("T", "S"), ocn.SeawaterBuoyancy(ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4))
    return (("T", "S"), ocn.SeawaterBuoyancy()) if zkind == "bounded" else (("T", "S"), None)


def _closure(ocn, zkind):
    if zkind == "amd":
        return ocn.AnisotropicMinimumDissipation(C=1 / 3, Cκ={"T": 1 / 3, "S": 1 / 12})
    return ocn.ScalarDiffusivity(ν=2e-3, κ={"T": 1e-3, "S": 5e-4}) if zkind == "bounded" else None


def _run_virtual_ranks(ocn, arch, R, size, nsteps, async_halos, zkind="periodic", thin_halos=True):
    import torch
    import host_orchestration as dist
    from loopback import LoopbackWorld
    world = LoopbackWorld(R, torch, arch)
    results, errors = [None] * R, []

    def worker(rank):
        try:
            ctx = world.context(rank)
            z, topo = _z_and_topology(ocn, zkind, size[2])
            grid = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
            model = dist.DistributedNonhydrostaticModel(grid=grid, tracers=("T", "S"), boundary_conditions=_bcs(ocn, zkind),
                                                        closure=_closure(ocn, zkind), buoyancy=_tracers_and_buoyancy(ocn, zkind)[1],
                                                        coriolis=ocn.FPlane(f=0.5) if zkind == "stretched" else None)
            model.async_halos = async_halos
            model.thin_halos = thin_halos
            flds = model.fields()
            vals = {n: analytic(n, *grid.global_nodes(f.loc)) for n, f in flds.items()}
            dist.set_model(model, **vals)
            dt = 0.1 * grid.local.Δxᶜᵃᵃ / 0.6
            for _ in range(nsteps):
                dist.time_step(model, dt)
            div = dist.max_abs_divergence(model)
            out = {n: f.parent() for n, f in flds.items()}
            out["p"] = model.pressure.parent()
            results[rank] = (out, div, model.time)
        except BaseException as e:          # noqa: BLE001
            errors.append((rank, repr(e)))
            world.barrier_obj.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    return results


@pytest.mark.parametrize("R,async_halos,size,zkind", [
    (2, False, (32, 16, 8), "periodic"), (2, True, (32, 16, 8), "periodic"), (4, True, (32, 16, 8), "periodic"),
    (4, True, (36, 12, 10), "periodic"),       # odd local Nx (9): padded column pair; Ny/2+1 = 7 modes over 4 ranks
    (8, True, (64, 8, 8), "periodic"),         # the rank count of one MI355X node: 8-point interface system of the x solve
    (3, True, (24, 9, 6), "periodic"),         # a rank count that is not a power of two (Ny must divide by it: the reference's validation)
    (8, True, (64, 8, 8), "stretched"),        # 8 ranks, transposing Fourier-tridiagonal solver
    (2, None, (512, 24, 16), "periodic"),      # local Nx = 256, automatic overlap choice: several tiles in every direction
    (2, True, (384, 8, 8), "periodic"),        # local Nx = 192: buffer strips one 64-lane tile wide (buffer_strip_width)
    (2, True, (32, 16, 8), "bounded"),         # z Bounded: distributed Fourier-tridiagonal solver
    (4, True, (32, 12, 10), "amd"),            # the whole configs[4] physics: stretched z, AMD (νₑ, κₑ evaluated in the x-halo columns),
                                               # linear seawater, wind stress / heat flux / bottom gradient / evaporation conditions
    (4, True, (28, 8, 12), "stretched"),       # stretched z, odd local Nx (7)
])
def test_virtual_ranks_match_single_gpu_and_oracle(ocn, oracle, arch, R, async_halos, size, zkind):
    _virtual_rank_case(ocn, oracle, arch, R, async_halos, size, zkind)


@pytest.mark.parametrize("R,size", [(2, (32, 16, 8)), (4, (36, 12, 10))])
def test_transposing_solver_still_matches(ocn, oracle, arch, R, size):
    """the all-to-all (transposed FFT) form of the distributed solver -- the reference's algorithm, kept behind the option
    dist_substructured = 0 and used whenever z is Bounded -- on the periodic cases the substructured solve takes by default"""
    ocn.set_option("dist_substructured", 0)
    try:
        _virtual_rank_case(ocn, oracle, arch, R, True, size, "periodic")
    finally:
        ocn.set_option("dist_substructured", 1)


def test_one_column_exchanges_equal_the_full_fills(ocn, arch):
    """compute_pressure_correction exchanges one column of u and of p (thin_halos) where the reference's generic fills move Hx
    columns of u, v, w and p: every field, halos included, is bit-identical after 3 steps (the deeper columns are never read
    before update_state! fills them again); p's own x halos are compared one column deep"""
    import ctypes as C
    import torch
    from oldoceananigans_jl_amd import _lib
    _lib.check(_lib.lib().ocn_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    thin = _run_virtual_ranks(ocn, arch, 2, (32, 16, 8), 3, True, "bounded", thin_halos=True)
    full = _run_virtual_ranks(ocn, arch, 2, (32, 16, 8), 3, True, "bounded", thin_halos=False)
    for (a, _, _), (b, _, _) in zip(thin, full):
        for name in a:
            if name == "p":
                assert np.array_equal(a[name][2:-2], b[name][2:-2])
            else:
                assert np.array_equal(a[name], b[name]), name


def _virtual_rank_case(ocn, oracle, arch, R, async_halos, size, zkind):
    import ctypes as C
    import torch
    from oldoceananigans_jl_amd import _lib
    nsteps = 3
    # the distributed path runs on torch's current stream
    _lib.check(_lib.lib().ocn_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    results = _run_virtual_ranks(ocn, arch, R, size, nsteps, async_halos, zkind)
    # single-GPU product model on the global grid
    z, topo = _z_and_topology(ocn, zkind, size[2])
    grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
    model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), boundary_conditions=_bcs(ocn, zkind), closure=_closure(ocn, zkind),
                                    buoyancy=_tracers_and_buoyancy(ocn, zkind)[1], coriolis=ocn.FPlane(f=0.5) if zkind == "stretched" else None)
    ocn.set_model(model, **{n: analytic(n, *grid.nodes(f.loc)) for n, f in model.fields().items()})
    dt = 0.1 * grid.Δxᶜᵃᵃ / 0.6
    for _ in range(nsteps):
        ocn.time_step(model, dt)
    glob = {n: f.parent() for n, f in model.fields().items()}
    glob["p"] = model.pressures.pNHS.parent()
    nxl = size[0] // R
    if zkind == "periodic":
        # third opinion: the serial oracle (tells which side is wrong if the two product paths ever disagree)
        from test_distributed_cpu import _serial
        om = _serial(oracle, size, nsteps)
        for name, cn in (("u", "u"), ("T", "c0"), ("p", "p")):
            ref = om.field(cn)
            assert rel_err(glob[name][3:-3, 3:-3, 3:-3], ref[3:-3, 3:-3, 3:-3]) < 1e-12, ("single-GPU model vs oracle", name)
    for r, (out, div, time) in enumerate(results):
        assert div < 5e-8 and time == model.clock.time
        for name, a in out.items():
            ref = glob[name][3 + r * nxl:3 + (r + 1) * nxl, 3:-3, 3:-3]
            scale = np.abs(glob[name]).max()
            err = np.abs(a[3:-3, 3:-3, 3:-3] - ref).max() / scale
            assert err <= 1e-12, (r, name, err, int(np.isnan(a).sum()), float(np.abs(a).max()), float(scale))
            if name != "p":   # x halos are COPIES of the neighbours' interior columns: bit-exact (==), like the reference's rank-id test
                west_rank, east_rank = results[(r - 1) % R][0][name], results[(r + 1) % R][0][name]
                assert np.array_equal(a[:3, 3:-3, 3:-3], west_rank[-6:-3, 3:-3, 3:-3]), (r, name, "west halo")
                assert np.array_equal(a[-3:, 3:-3, 3:-3], east_rank[3:6, 3:-3, 3:-3]), (r, name, "east halo")


def _in_virtual_ranks(arch, R, fn):
    """run fn(ctx, dist) on R virtual ranks (threads) and return the per-rank results"""
    import torch
    import host_orchestration as dist
    from loopback import LoopbackWorld
    world = LoopbackWorld(R, torch, arch)
    results, errors = [None] * R, []

    def worker(rank):
        try:
            results[rank] = fn(world.context(rank), dist)
        except BaseException as e:          # noqa: BLE001
            errors.append((rank, repr(e)))
            world.barrier_obj.abort()
    threads = [threading.Thread(target=worker, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    return results


@pytest.mark.parametrize("R", [2, 4])
def test_halo_exchange_with_rank_ids_is_exact(ocn, arch, R):
    """test/test_distributed_models.jl:334-404: every rank fills its fields with its rank id; after fill_halo_regions! the west /
    east halos hold the neighbours' ids exactly (`==`), corners included, y / z halos the own id"""
    import ctypes as C
    import torch
    from oldoceananigans_jl_amd import _lib
    _lib.check(_lib.lib().ocn_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def fn(ctx, dist):
        grid = dist.DistributedRectilinearGrid(ctx, size=(8 * R, 8, 5), x=(0.0, 1.0), y=(0.0, 1.0), z=(-1.0, 0.0),
                                               topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
        model = dist.DistributedNonhydrostaticModel(grid=grid, tracers=("T", "S"))
        b = model.backend
        for n, f in enumerate(b.U):
            f.set_parent(np.full(f.shape, float(100 * n + ctx.rank)))
        dist.fill_halo_regions(model, b.U, fill_open_bcs=False)
        return [f.parent() for f in b.U]
    out = _in_virtual_ranks(arch, R, fn)
    for r in range(R):
        for n, a in enumerate(out[r]):
            west, east = 100 * n + (r - 1) % R, 100 * n + (r + 1) % R
            assert np.all(a[:3, :, :] == west) and np.all(a[-3:, :, :] == east), (r, n)       # whole y / z extent: corners ride along
            assert np.all(a[3:-3, :, :] == 100 * n + r)


@pytest.mark.parametrize("R,size", [(2, (16, 8, 6)), (4, (36, 12, 10))])
def test_transposes_round_trip(ocn, arch, R, size):
    """test/test_distributed_transpose.jl:13-54: y -> x -> y transposes return the original field. Here: the transposing solver's
    forward stage (local transform, separation, pack), all-to-all, all-to-all back, backward stage (rebuild, inverse transform) with
    NO solve in between returns the source term times Ny Nz -- the index algebra of pack / unpack / Hermitian rebuild is the identity"""
    import ctypes as C
    import torch
    from oldoceananigans_jl_amd import _lib
    _lib.check(_lib.lib().ocn_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ocn.set_option("dist_substructured", 0)

    def fn(ctx, dist):
        grid = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=(0.0, 1.0))
        model = dist.DistributedNonhydrostaticModel(grid=grid, tracers=())
        b = model.backend
        rng = np.random.default_rng(ctx.rank)
        for f in b.U[:3]:
            f.set_parent(rng.standard_normal(f.shape))
        dist.fill_halo_regions(model, b.U[:3])
        b.source_term()
        g = grid.local
        u, v, w = (f.parent() for f in b.U[:3])
        div = ((u[4:-2, 3:-3, 3:-3] - u[3:-3, 3:-3, 3:-3]) / g.Δxᶜᵃᵃ + (v[3:-3, 4:-2, 3:-3] - v[3:-3, 3:-3, 3:-3]) / g.Δyᵃᶜᵃ +
               (w[3:-3, 3:-3, 4:-2] - w[3:-3, 3:-3, 3:-3]) / g.Δzᵃᵃᶜ[3])
        b.poisson_forward_yz()
        ctx.all_to_all(b.recv, b.send)          # transpose_y_to_x!
        b.send.copy_(b.recv)                    # (no x stage)
        ctx.all_to_all(b.recv, b.send)          # transpose_x_to_y!
        b.poisson_backward_yz()
        return b.p.parent()[3:-3, 3:-3, 3:-3] / (size[1] * size[2]), div
    try:
        out = _in_virtual_ranks(arch, R, fn)
    finally:
        ocn.set_option("dist_substructured", 1)
    for back, div in out:
        assert np.abs(back - div).max() < 1e-12 * np.abs(div).max()


@pytest.mark.parametrize("R,size,zkind,substructured", [(2, (32, 16, 8), "periodic", 1), (4, (32, 16, 8), "periodic", 1),
                                                        (2, (32, 16, 8), "periodic", 0), (2, (32, 16, 12), "stretched", 1)])
def test_separate_processes_match_single_gpu(ocn, arch, tmp_path, R, size, zkind, substructured):
    """the N > 1 product path as the driver launches it -- torch.distributed.run, one process per rank, DeviceBackend, rank-local
    coordinates from the environment -- except that the R processes share this box's one card and stage their collectives
    through the host (HostStagedContext). Fields after 3 RK3 steps against the single-GPU model on the global grid."""
    import subprocess
    import sys
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    nsteps = 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={R}", "--master-addr", "127.0.0.1",
           "--master-port", str(29560 + R + 10 * substructured + (20 if zkind == "stretched" else 0)), os.path.join(here, "gpu_dist_worker.py"), str(tmp_path),
           str(size[0]), str(size[1]), str(size[2]), str(nsteps), zkind, str(substructured)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    z, topo = _z_and_topology(ocn, zkind, size[2])
    grid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
    model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
    ocn.set_model(model, **{n: analytic(n, *grid.nodes(f.loc)) for n, f in model.fields().items()})
    dt = 0.1 * grid.Δxᶜᵃᵃ / 0.6
    for _ in range(nsteps):
        ocn.time_step(model, dt)
    glob = {n: f.parent() for n, f in model.fields().items()}
    glob["p"] = model.pressures.pNHS.parent()
    nxl = size[0] // R
    for r in range(R):
        out = np.load(tmp_path / f"rank{r}.npz")
        assert float(out["div"]) < 5e-8 and float(out["time"]) == model.clock.time
        for name, ref in glob.items():
            a = out[name]
            err = np.abs(a[3:-3, 3:-3, 3:-3] - ref[3 + r * nxl:3 + (r + 1) * nxl, 3:-3, 3:-3]).max() / np.abs(ref).max()
            assert err <= 1e-12, (r, name, err)


@pytest.mark.parametrize("size,zkind,substructured", [((32, 16, 8), "periodic", 1), ((32, 16, 8), "periodic", 0), ((32, 12, 10), "stretched", 1),
                                                      ((384, 8, 8), "periodic", 1)])
def test_self_loop_rank_equals_single_gpu(ocn, arch, size, zkind, substructured):
    """SelfLoopContext: ONE rank that is its own neighbour runs the whole partitioned code path (FullyConnected x, pack / exchange /
    unpack, strips, thin exchanges, gathered interface solve) with device copies; the fields equal the single-GPU model's"""
    import ctypes as C
    import torch
    from oldoceananigans_jl_amd import _lib
    import host_orchestration as dist
    _lib.check(_lib.lib().ocn_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ocn.set_option("dist_substructured", substructured)
    try:
        ctx = dist.SelfLoopContext(0, 1, torch.device("cuda", 0), torch, None, arch)
        z, topo = _z_and_topology(ocn, zkind, size[2])
        grid = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
        assert grid.local.topology[0] is ocn.FullyConnected
        model = dist.DistributedNonhydrostaticModel(grid=grid, tracers=("T", "S"), boundary_conditions=_bcs(ocn, zkind),
                                                    coriolis=ocn.FPlane(f=0.5) if zkind == "stretched" else None)
        flds = model.fields()
        dist.set_model(model, **{n: analytic(n, *grid.global_nodes(f.loc)) for n, f in flds.items()})
        dt = 0.1 * grid.local.Δxᶜᵃᵃ / 0.6
        for _ in range(3):
            dist.time_step(model, dt)
        out = {n: f.parent() for n, f in flds.items()}
        out["p"] = model.pressure.parent()
        model.backend.close()
    finally:
        ocn.set_option("dist_substructured", 1)
    sgrid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
    smodel = ocn.NonhydrostaticModel(grid=sgrid, tracers=("T", "S"), boundary_conditions=_bcs(ocn, zkind),
                                     coriolis=ocn.FPlane(f=0.5) if zkind == "stretched" else None)
    ocn.set_model(smodel, **{n: analytic(n, *sgrid.nodes(f.loc)) for n, f in smodel.fields().items()})
    for _ in range(3):
        ocn.time_step(smodel, dt)
    ref = {n: f.parent() for n, f in smodel.fields().items()}
    ref["p"] = smodel.pressures.pNHS.parent()
    for name, a in out.items():
        err = np.abs(a[3:-3, 3:-3, 3:-3] - ref[name][3:-3, 3:-3, 3:-3]).max() / np.abs(ref[name]).max()
        assert err <= 1e-12, (name, err)


def test_substructured_solver_layouts(ocn, arch):
    """the three local layouts of the substructured x solve agree: z-fastest with the 2-D (y, z) real plan, z-fastest with 1-D plans
    (what rocFFT leaves when it refuses the interleaved-batch 2-D layout), paired real columns (option dist_zfirst = 0)"""
    import ctypes as C
    import torch
    from oldoceananigans_jl_amd import _lib
    import host_orchestration as dist
    _lib.check(_lib.lib().ocn_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    outs, layouts = [], []
    for size, zfirst in (((256, 24, 16), 1), ((256, 24, 16), 0), ((64, 24, 16), 1), ((64, 24, 16), 0)):
        ocn.set_option("dist_zfirst", zfirst)
        try:
            ctx = dist.SelfLoopContext(0, 1, torch.device("cuda", 0), torch, None, arch)
            grid = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=(0.0, 1.0))
            model = dist.DistributedNonhydrostaticModel(grid=grid, tracers=("T", "S"))
            lay = C.c_int()
            _lib.check(_lib.lib().ocn_dist_poisson_layout(model.backend.solver, C.byref(lay)))
            layouts.append(lay.value)
            dist.set_model(model, **{n: analytic(n, *grid.global_nodes(f.loc)) for n, f in model.fields().items()})
            for _ in range(2):
                dist.time_step(model, 0.1 * grid.local.Δxᶜᵃᵃ / 0.6)
            outs.append({n: f.parent() for n, f in model.fields().items()} | {"p": model.pressure.parent()})
            model.backend.close()
        finally:
            ocn.set_option("dist_zfirst", 1)
    assert layouts[1] == 0 and layouts[3] == 0 and layouts[0] in (1, 2, 3) and layouts[2] in (1, 2, 3)
    for a, b in ((outs[0], outs[1]), (outs[2], outs[3])):
        for name in a:
            err = np.abs(a[name][3:-3, 3:-3, 3:-3] - b[name][3:-3, 3:-3, 3:-3]).max() / np.abs(b[name]).max()
            assert err <= 1e-12, (name, err)


def test_simulation_drives_a_partitioned_model(ocn, arch):
    """Simulation(model) with the library's partitioned model (a one-rank self-loop: the N > 1 code path): run! with a stop iteration and a
    TimeInterval callback, reset!, against the same Simulation of the single-GPU model"""
    import ctypes as C
    from oldoceananigans_jl_amd import _lib, distributed as dist
    _lib.check(_lib.lib().ocn_own_stream())
    size = (32, 16, 8)
    z, topo = _z_and_topology(ocn, "periodic", size[2])

    def drive(model, nodes_of):
        ocn.set_model(model, **{n: analytic(n, *nodes_of(f.loc)) for n, f in model.fields().items()})
        dt = 0.1 * (2.0 / size[0]) / 0.6
        sim = ocn.Simulation(model, Δt=dt, stop_iteration=5)
        hits = []
        sim.callbacks["probe"] = ocn.Callback(lambda s: hits.append((s.model.clock.iteration, s.model.clock.time)), ocn.TimeInterval(2.5 * dt))
        ocn.run(sim)
        assert model.clock.iteration == 5
        first = ({n: f.parent() for n, f in model.fields().items()}, list(hits), model.clock.time)
        ocn.reset(sim)
        assert model.clock.iteration == 0 and model.clock.time == 0.0
        return first

    uid = C.create_string_buffer(128)
    _lib.check(_lib.lib().ocn_dist_unique_id(uid))
    ctx = dist.Distributed.rccl(arch, uid, 1, 0, self_loop=True)
    grid = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
    lib_model = dist.LibraryDistributedModel(grid=grid, tracers=("T", "S"))
    out_lib, hits_lib, t_lib = drive(lib_model, grid.global_nodes)
    lib_model.close()
    ctx.close()
    sgrid = ocn.RectilinearGrid(arch, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
    smodel = ocn.NonhydrostaticModel(grid=sgrid, tracers=("T", "S"))
    out_ref, hits_ref, t_ref = drive(smodel, sgrid.nodes)
    assert t_lib == t_ref and [h[0] for h in hits_lib] == [h[0] for h in hits_ref] and len(hits_ref) >= 2
    assert all(abs(a[1] - b[1]) < 1e-15 for a, b in zip(hits_lib, hits_ref))
    for n in out_ref:
        err = np.abs(out_lib[n][3:-3, 3:-3, 3:-3] - out_ref[n][3:-3, 3:-3, 3:-3]).max() / np.abs(out_ref[n]).max()
        assert err <= 1e-12, (n, err)


def test_cell_diffusion_timescale_with_eddy_coefficients(ocn, arch):
    """cell_diffusion_timescale for AnisotropicMinimumDissipation: Δ² / max(νₑ, κₑ) over the diffusivity fields
    (Simulations' diffusive CFL; the closure has no constant ν, κ) -- the wizard accepts diffusive_cfl on the configs[4] physics"""
    grid = ocn.RectilinearGrid(arch, size=(16, 12, 10), x=(0, 1), y=(0, 1), z=(-1, 0), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
    model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), closure=ocn.AnisotropicMinimumDissipation())
    rng = np.random.default_rng(2)
    ocn.set_model(model, **{n: rng.standard_normal(grid.interior_size(f.loc)) for n, f in model.fields().items()})
    D = model.diffusivity_fields
    biggest = max(float(D[0].parent().max()), *(float(k.parent().max()) for k in D[1]))
    assert biggest > 0
    delta = min(grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, float(np.min(grid.Δzᵃᵃᶜ[3:-4])))
    assert ocn.cell_diffusion_timescale(model) == delta ** 2 / biggest
    wizard = ocn.TimeStepWizard(cfl=0.5, diffusive_cfl=0.1)
    dt = ocn.new_time_step(1e-3, wizard, model)
    assert 0 < dt <= 1.1e-3
