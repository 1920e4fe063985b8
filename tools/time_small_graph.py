"""ms/step of small single-GPU models with and without the hipGraph replay of the RK3 step (GPU box): python tools/time_small_graph.py 32 64 128"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state
arch = ocn.GPU(0)
for N in [int(v) for v in sys.argv[1:]]:
    for graph in (0, 1):
        grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
        model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
        model.set_option("use_graph", graph)
        ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 1234))
        dt = 0.1 / N / 0.6
        for _ in range(5): ocn.time_step(model, dt)
        steps = 200 if N <= 64 else 50
        ocn.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): ocn.time_step(model, dt)
        ocn.synchronize()
        print(f"{N}^3 use_graph={graph}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step", flush=True)
        model.close()
