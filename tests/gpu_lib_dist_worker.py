"""worker of the multi-GPU RCCL test (tests/test_gpu_dist_library.py::test_library_rccl_separate_processes, launched through
torch.distributed.run, one rank per GPU): the product path -- library-owned communicator, partitioned step inside the library --
dumps this rank's slab. No torch in this process -- unless OCN_TEST_HOST_STAGED=1, the one-GPU rehearsal of the same path: all ranks on card
0, the library's orchestration over a host-staged gloo transport."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

STAGED = os.environ.get("OCN_TEST_HOST_STAGED") == "1"     # one-GPU rehearsal: all ranks on card 0, collectives over gloo (tests/host_staged.py)
if STAGED:
    import torch  # noqa: E402,F401  -- before the library (tests/conftest.py explains)
import oldoceananigans_jl_amd as ocn  # noqa: E402
from oldoceananigans_jl_amd import distributed as dist  # noqa: E402
from dist_worker import analytic  # noqa: E402
from helpers import tanh_faces  # noqa: E402


def main():
    outdir, nsteps, zkind = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    size = tuple(int(x) for x in sys.argv[4:7])
    options = dict(kv.split("=") for kv in sys.argv[7:])
    if STAGED:
        import torch.distributed as td
        from host_staged import HostStagedCollectives
        from oldoceananigans_jl_amd import _lib
        td.init_process_group("gloo")
        rank, world = td.get_rank(), td.get_world_size()
        arch = ocn.GPU(0)
        ctx = dist.Distributed.transport(arch, HostStagedCollectives(torch, td, _lib.lib(), rank, world), world, rank)
        for key in ("dist_substructured",):
            if key in options:
                ocn.set_option(key, int(options.pop(key)))
    else:
        ctx = dist.Distributed.from_environment()
    if zkind == "periodic":
        z, topo = (0.0, 1.0), (ocn.Periodic, ocn.Periodic, ocn.Periodic)
    else:
        z, topo = tanh_faces(size[2]), (ocn.Periodic, ocn.Periodic, ocn.Bounded)
    grid = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
    model = dist.LibraryDistributedModel(grid=grid, tracers=("T", "S"))
    for k, v in options.items():
        model.set_option(k, int(v))
    ocn.set_model(model, **{n: analytic(n, *grid.global_nodes(f.loc)) for n, f in model.fields().items()})
    dt = 0.1 * (2.0 / size[0]) / 0.6
    for _ in range(nsteps):
        ocn.time_step(model, dt)
    div = ocn.max_abs_divergence(model)
    out = {n: f.parent() for n, f in model.fields().items()}
    out["p"] = model.pressures.pNHS.parent()
    np.savez(os.path.join(outdir, f"rank{ctx.rank}.npz"), div=div, time=model.clock.time, **out)
    ctx.barrier()
    model.close()
    ctx.close()


if __name__ == "__main__":
    main()
