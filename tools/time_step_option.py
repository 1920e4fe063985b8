"""A/B of one library option on the whole time-step (GPU box): python tools/time_step_option.py <option> [N] -> ms/step with the option 1 and 0"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state
opt = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
arch = ocn.GPU(0)
for rep in range(2):
    for val in (1, 0):
        ocn.set_option(opt, val)
        grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
        model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
        ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 1234))
        dt = 0.1 / N / 0.6
        for _ in range(5): ocn.time_step(model, dt)
        ocn.synchronize(); t0 = time.perf_counter()
        for _ in range(30): ocn.time_step(model, dt)
        ocn.synchronize()
        print(f"{opt} = {val}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
        model.close()
ocn.set_option(opt, 1)
