"""stability check on the GPU box: many RK3 steps at CFL ~ 0.1-0.3 from the smooth benchmark state; prints energy, divergence, NaN state"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
arch = ocn.GPU(0)
grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 1234))
wizard = ocn.TimeStepWizard(cfl=0.3, max_change=1.1)
dt = 0.1 / N / 0.6
t0 = time.perf_counter()
for n in range(steps):
    if n % 100 == 0:
        dt = ocn.new_time_step(dt, wizard, model)
        u, T = model.velocities.u.interior(), model.tracers.T.interior()
        ke = float((u ** 2).mean())
        print(f"step {n:5d} t = {model.clock.time:.4f} dt = {dt:.2e} <u^2> = {ke:.6f} T in [{T.min():.4f}, {T.max():.4f}] "
              f"div = {ocn.max_abs_divergence(model):.2e} nan = {ocn.hasnan(model)}", flush=True)
    ocn.time_step(model, dt)
print(f"{steps} steps in {time.perf_counter() - t0:.1f} s; final nan = {ocn.hasnan(model)}")
