"""ms/step of the single-GPU model at N^3 (GPU box): python tools/time_step_size.py N [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state
N = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
arch = ocn.GPU(0)
grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 1234))
dt = 0.1 / N / 0.6
for _ in range(3): ocn.time_step(model, dt)
for rep in range(2):
    ocn.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): ocn.time_step(model, dt)
    ocn.synchronize()
    print(f"{N}^3: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step, max|div u| {ocn.max_abs_divergence(model):.2e}", flush=True)
