"""worker of the world_size-2 gloo test (launched by tests/test_distributed_cpu.py through torch.distributed.run):
runs the product's distributed orchestration on the test-only CPU backend and dumps this rank's slab."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oldoceananigans_jl_amd as ocn  # noqa: E402
import host_orchestration as dist  # noqa: E402
from cpu_backend import CpuBackend, OracleLocalGrid  # noqa: E402
from helpers import smooth_state  # noqa: E402


def main():
    outdir, nsteps, async_halos = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    size = tuple(int(x) for x in sys.argv[4:7])
    ctx = dist.init_process_group(0, backend="gloo")
    grid = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=(0.0, 1.0),
                                           make_local_grid=lambda *a: OracleLocalGrid(*a))
    backend = CpuBackend(ctx, grid, 2, ocn)
    model = dist.DistributedNonhydrostaticModel(grid=grid, tracers=("T", "S"), backend=backend)
    model.async_halos = bool(async_halos)
    if async_halos > 1:
        model.strip_width = async_halos          # wider-than-Hx buffer strips (the product uses one 64-lane tile at 256^3)
    flds = model.fields()
    vals = smooth_state({n: grid.global_nodes(f.loc) for n, f in flds.items()}, 1234 + ctx.rank)
    # deterministic global noise: regenerate from the global coordinates instead of a per-rank rng
    vals = {n: v - 0 for n, v in vals.items()}
    for n in vals:
        x, y, z = grid.global_nodes(flds[n].loc)
        vals[n] = analytic(n, x, y, z)
    dist.set_model(model, **vals)
    dt = 0.1 * grid.local.Δxᶜᵃᵃ / 0.6
    for _ in range(nsteps):
        dist.time_step(model, dt)
    div = dist.max_abs_divergence(model)
    out = {n: np.array(f.parent(), copy=True) for n, f in flds.items()}
    out["p"] = np.array(backend.p.parent(), copy=True)
    # the collective of the substructured pressure solve (DistributedContext.all_gather): rank r's piece lands at slot r everywhere
    import torch
    piece = torch.full((5,), float(ctx.rank + 1), dtype=torch.float64)
    gathered = torch.zeros(5 * ctx.world, dtype=torch.float64)
    ctx.all_gather(gathered, piece)
    gather_ok = bool(all(torch.all(gathered[r * 5:(r + 1) * 5] == r + 1) for r in range(ctx.world)))
    # test/test_distributed_models.jl:334-404: fields filled with the rank id; after the exchange the x halos hold the neighbours' ids
    for n, f in enumerate(backend.U):
        f.a[...] = 100 * n + ctx.rank
    dist.fill_halo_regions(model, backend.U, fill_open_bcs=False)
    west, east = (ctx.rank - 1) % ctx.world, (ctx.rank + 1) % ctx.world
    ids_ok = all(bool(np.all(f.a[:3] == 100 * n + west) and np.all(f.a[-3:] == 100 * n + east) and np.all(f.a[3:-3] == 100 * n + ctx.rank))
                 for n, f in enumerate(backend.U))
    np.savez(os.path.join(outdir, f"rank{ctx.rank}.npz"), div=div, time=model.time, iteration=model.iteration, gather_ok=gather_ok,
             ids_ok=ids_ok, **out)
    ctx.barrier()


def analytic(name, x, y, z):
    two_pi = 2 * np.pi
    if name == "u":
        return 0.5 * np.sin(np.pi * x) * np.cos(two_pi * y) * np.cos(two_pi * z) + 0.05 * np.sin(3 * np.pi * x + 1.0) + 0 * y * z
    if name == "v":
        return -0.5 * np.cos(np.pi * x) * np.sin(two_pi * y) * np.cos(two_pi * z) + 0.05 * np.cos(two_pi * z + 0.3) + 0 * x * y
    if name == "w":
        return 0.1 * np.cos(np.pi * x) * np.cos(two_pi * y) * np.sin(two_pi * z) + 0.05 * np.sin(two_pi * y + 0.7) + 0 * x * z
    if name == "T":
        return np.exp(-((x - 1.0) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) / 0.05)
    # one term per direction (see helpers.smooth_state): a tracer that is exactly uniform along a direction next to an offset of 35 makes the
    # WENO smoothness indicators there pure round-off, and ulp-level differences of the metrics (a local slab's Δx is L_local / N_local) grow
    # to 1e-12 in three steps -- conditioning of the scheme, not of an implementation
    return 35 + np.sin(np.pi * x) * np.cos(two_pi * y) + 0.2 * np.cos(two_pi * z) + 0.3 * np.sin(two_pi * y) + 0.25 * np.cos(np.pi * x)


if __name__ == "__main__":
    main()
