"""tendency evaluation time at 256^3 for 0, 1, 2, 3 tracers (how the fused kernel scales with the number of reconstructions)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oldoceananigans_jl_amd as ocn
from oldoceananigans_jl_amd import kernels
from helpers import smooth_state
N = 256
arch = ocn.GPU(0)
grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
for names in ((), ("T",), ("T", "S"), ("T", "S", "C")):
    model = ocn.NonhydrostaticModel(grid=grid, tracers=names)
    flds = model.fields()
    st = smooth_state({n: grid.nodes(f.loc) for n, f in flds.items() if n != "C"}, 1234)
    if "C" in flds:
        st["C"] = st["T"]
    ocn.set_model(model, **st)
    F = list(flds.values()); G = [model.tendency(n) for n in flds]
    for _ in range(5):
        kernels.compute_tendencies(grid, F[0], F[1], F[2], F[3:], G[0], G[1], G[2], G[3:], None)
    ocn.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        kernels.compute_tendencies(grid, F[0], F[1], F[2], F[3:], G[0], G[1], G[2], G[3:], None)
    ocn.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 20
    nrec = 9 + 3 * len(names)
    print(f"{len(names)} tracers: {ms:.3f} ms  ({nrec} reconstructions/cell -> {1e3 * ms / nrec:.1f} us per reconstruction-sweep)", flush=True)
    del model
