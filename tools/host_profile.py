"""cProfile of the Python-side orchestration of the distributed time step (world = 1) -- run on the GPU box"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch  # noqa
import oldoceananigans_jl_amd as ocn
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_orchestration as dist
from bench import initial_state
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = dist.init_process_group(0)
grid = dist.DistributedRectilinearGrid(ctx, size=(N, N, N), extent=(1.0, 1.0, 1.0))
model = dist.DistributedNonhydrostaticModel(grid=grid, advection=ocn.WENO(), tracers=("T", "S"))
dist.set_model(model, **dist.local_initial_state(model, initial_state))
dt = 0.1 / N / 0.6
for _ in range(3):
    dist.time_step(model, dt)
ocn.synchronize()
# host-only cost: issue 10 steps without waiting, time the issue loop, then the drain
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    dist.time_step(model, dt)
pr.disable()
t1 = time.perf_counter()
ocn.synchronize()
t2 = time.perf_counter()
print(f"issue {1e3*(t1-t0)/10:.2f} ms/step, drain {1e3*(t2-t1):.2f} ms total")
model.set_option("profile", 1)
ocn.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dist.time_step(model, dt)
t1 = time.perf_counter()
ocn.synchronize()
t2 = time.perf_counter()
print(f"with event profiling: issue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
model.set_option("profile", 0)
ctx.barrier(); ocn.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dist.time_step(model, dt)
ocn.synchronize(); ctx.barrier()
t2 = time.perf_counter()
print(f"barrier-bracketed: total {1e3*(t2-t0)/10:.2f} ms/step")
pstats.Stats(pr).sort_stats("cumulative").print_stats(3)
