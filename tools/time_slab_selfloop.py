"""ms/step of ONE x-slab of a partitioned model run through the in-library partitioned step by a one-rank RCCL communicator that is its
own neighbour (GPU box): python tools/time_slab_selfloop.py Nx Ny Nz [bounded] [steps] -- e.g. 64 512 512 (configs[3]), 128 1024 256 bounded"""
import ctypes as C, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oldoceananigans_jl_amd as ocn
from oldoceananigans_jl_amd import _lib, distributed as dist
from helpers import smooth_state, tanh_faces
shape = tuple(int(v) for v in sys.argv[1:4]); bounded = len(sys.argv) > 4 and sys.argv[4] == "bounded"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
arch = ocn.GPU(0)
for kv in filter(None, os.environ.get("OCN_SET_OPTIONS", "").split(",")):      # library options for A/B runs: OCN_SET_OPTIONS=key=value,...
    ocn.set_option(kv.split("=")[0], int(kv.split("=")[1]))
uid = C.create_string_buffer(128)
_lib.check(_lib.lib().ocn_dist_unique_id(uid))
ctx = dist.Distributed.rccl(arch, uid, 1, 0, self_loop=True)
topo = (ocn.Periodic, ocn.Periodic, ocn.Bounded if bounded else ocn.Periodic)
grid = dist.DistributedRectilinearGrid(ctx, size=shape, x=(0.0, 1.0), y=(0.0, 1.0), z=tanh_faces(shape[2]) if bounded else (0.0, 1.0), topology=topo)
model = dist.LibraryDistributedModel(grid=grid, tracers=("T", "S"))
vals = smooth_state({n: grid.global_nodes(f.loc) for n, f in model.fields().items()}, 99)
vals["S"] = vals["S"] - 35.0
ocn.set_model(model, **vals)
dt = 0.1 / max(shape) / 0.6
for _ in range(3): ocn.time_step(model, dt)
for rep in range(2):
    ocn.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): ocn.time_step(model, dt)
    ocn.synchronize()
    print(f"slab {shape} {'PPB' if bounded else 'PPP'} (self-loop): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step, max|div u| {ocn.max_abs_divergence(model):.2e}", flush=True)
model.close(); ctx.close()
