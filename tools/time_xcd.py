"""tendency evaluation at 256^3 with / without the XCD-aware tile order (GPU box)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oldoceananigans_jl_amd as ocn
from oldoceananigans_jl_amd import kernels
from helpers import smooth_state
N = 256
arch = ocn.GPU(0)
grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
flds = model.fields()
ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in flds.items()}, 1234))
F = list(flds.values()); G = [model.tendency(n) for n in flds]
ref = None
for rep in range(3):
    for xcd in (0, 1):
        ocn.set_option("fused_xcd", xcd)
        for _ in range(5):
            kernels.compute_tendencies(grid, F[0], F[1], F[2], F[3:], G[0], G[1], G[2], G[3:], None)
        ocn.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            kernels.compute_tendencies(grid, F[0], F[1], F[2], F[3:], G[0], G[1], G[2], G[3:], None)
        ocn.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 20
        out = [g.parent() for g in G]
        if ref is None: ref = out
        same = all(np.array_equal(a, b) for a, b in zip(out, ref))
        print(f"xcd swizzle {xcd}: {ms:.3f} ms  identical {same}", flush=True)
