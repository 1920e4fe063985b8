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
    if zkind != "stretched":
        return None
    F = ocn.FieldBoundaryConditions
    return {"u": F(top=ocn.FluxBoundaryCondition(-2e-3), bottom=ocn.ValueBoundaryCondition(0.0)),
            "T": F(top=ocn.FluxBoundaryCondition(5e-3), bottom=ocn.GradientBoundaryCondition(0.4))}


def _tracers_and_buoyancy(ocn, zkind):
    return (("T", "S"), ocn.SeawaterBuoyancy()) if zkind == "bounded" else (("T", "S"), None)


def _closure(ocn, zkind):
    return ocn.ScalarDiffusivity(ν=2e-3, κ={"T": 1e-3, "S": 5e-4}) if zkind == "bounded" else None


def _run_virtual_ranks(ocn, arch, R, size, nsteps, async_halos, zkind="periodic"):
    import torch
    from oldoceananigans_jl_amd import distributed as dist
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
            flds = model.fields()
            vals = {n: analytic(n, *grid.local.nodes(f.loc)) for n, f in flds.items()}
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
    (2, True, (32, 16, 8), "bounded"),         # z Bounded: distributed Fourier-tridiagonal solver
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
            if name != "p":   # x halos hold exact copies of the neighbours' interiors (bit-exact exchange)
                lo = 3 + r * nxl - 3
                west = glob[name][lo:lo + 3, 3:-3, 3:-3] if r > 0 else glob[name][size[0]:size[0] + 3, 3:-3, 3:-3]
                assert rel_err(a[:3, 3:-3, 3:-3], west) <= 1e-12
