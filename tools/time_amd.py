"""one line: the eddy-diffusivity kernel at 256 x 256 x 128 (configs[4] physics), ms per launch (GPU box)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state, tanh_faces
arch = ocn.GPU(0)
N = (256, 256, 128)
grid = ocn.RectilinearGrid(arch, size=N, x=(0, 1), y=(0, 1), z=tanh_faces(N[2]), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
closure = ocn.AnisotropicMinimumDissipation()
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), closure=closure)
ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 1234))
flds = list(model.fields().values())
nu, ka = ocn.CenterField(grid), [ocn.CenterField(grid), ocn.CenterField(grid)]
out = []
for rep in (0, 1):
    for _ in range(3):
        ocn.kernels.compute_amd_diffusivities(grid, closure, ("T", "S"), flds, nu, ka)
    ocn.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ocn.kernels.compute_amd_diffusivities(grid, closure, ("T", "S"), flds, nu, ka)
    ocn.synchronize()
    out.append((time.perf_counter() - t0) / 20 * 1e3)
print("amd diffusivities 256x256x128: %.3f / %.3f ms per launch" % tuple(out), flush=True)
