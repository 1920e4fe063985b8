import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state
arch = ocn.GPU(0)
for size in ((16, 512, 8), (16, 8, 512), (24, 1024, 8)):
    outs = []
    for split in (1, 0):
        ocn.set_option("split_solve", split)
        grid = ocn.RectilinearGrid(arch, size=size, extent=(1, 1, 1))
        model = ocn.NonhydrostaticModel(grid=grid, tracers=("T",))
        ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 5))
        for _ in range(2):
            ocn.time_step(model, 1e-4)
        outs.append(model.velocities.u.parent())
        print(size, "split", split, "div", ocn.max_abs_divergence(model), flush=True)
        del model
    print("   max diff", np.abs(outs[0] - outs[1]).max())
ocn.set_option("split_solve", 1)
