"""ms per RK3 time-step of the single-GPU model at small sizes, with and without the captured time-step graph (GPU box)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state
arch = ocn.GPU(0)
for N in (16, 32, 64, 128, 256):
    grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
    out = []
    for use_graph in (0, 1):
        model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
        model.set_option("use_graph", use_graph)
        ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 1234))
        dt = 0.1 / N / 0.6
        steps = 300 if N <= 64 else (100 if N == 128 else 20)
        for _ in range(10):
            ocn.time_step(model, dt)
        ocn.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ocn.time_step(model, dt)
        ocn.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / steps)
        assert model.get_option("graph_failures") == 0
    print(f"{N:4d}^3: launches {out[0]:.3f} ms/step   graph replay {out[1]:.3f} ms/step   x{out[0] / out[1]:.2f}", flush=True)
