"""worker of test_gpu_distributed.py::test_separate_processes_*: one REAL process per rank, all on card 0, collectives staged
through the host over gloo (distributed.HostStagedContext -- RCCL refuses two ranks on one device). Launched by
torch.distributed.run; writes this rank's fields to <outdir>/rank<r>.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    outdir, nx, ny, nz, nsteps, zkind, substructured = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), \
        int(sys.argv[5]), sys.argv[6], int(sys.argv[7])
    import torch  # noqa: F401  (before the library)
    import oldoceananigans_jl_amd as ocn
    import host_orchestration as dist
    from dist_worker import analytic
    from helpers import tanh_faces
    ctx = dist.init_process_group(int(os.environ.get("LOCAL_RANK", "0")), rehearse_on_one_gpu=True)
    ocn.set_option("dist_substructured", substructured)
    if zkind == "periodic":
        z, topo = (0.0, 1.0), (ocn.Periodic, ocn.Periodic, ocn.Periodic)
    else:
        z, topo = tanh_faces(nz), (ocn.Periodic, ocn.Periodic, ocn.Bounded)
    grid = dist.DistributedRectilinearGrid(ctx, size=(nx, ny, nz), x=(0.0, 2.0), y=(0.0, 1.0), z=z, topology=topo)
    model = dist.DistributedNonhydrostaticModel(grid=grid, tracers=("T", "S"))
    flds = model.fields()
    dist.set_model(model, **{n: analytic(n, *grid.global_nodes(f.loc)) for n, f in flds.items()})
    dt = 0.1 * grid.local.Δxᶜᵃᵃ / 0.6
    for _ in range(nsteps):
        dist.time_step(model, dt)
    out = {n: f.parent() for n, f in flds.items()}
    out["p"] = model.pressure.parent()
    out["div"] = np.array(dist.max_abs_divergence(model))
    out["time"] = np.array(model.time)
    np.savez(os.path.join(outdir, f"rank{ctx.rank}.npz"), **out)
    ctx.barrier()
    model.backend.close()
    ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
