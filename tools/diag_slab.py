"""diagnostic: partitioned (self-loop over RCCL) vs single-GPU on one slab, per-field error after every step"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oldoceananigans_jl_amd as ocn
from oldoceananigans_jl_amd import _lib, distributed as dist
from helpers import smooth_state
size = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 128, 128)
offset = float(sys.argv[4]) if len(sys.argv) > 4 else None
arch = ocn.GPU(0)
uid = C.create_string_buffer(128)
_lib.check(_lib.lib().ocn_dist_unique_id(uid))
ctx = dist.Distributed.rccl(arch, uid, 1, 0, self_loop=True)
topo = (ocn.Periodic,) * 3
g1 = dist.DistributedRectilinearGrid(ctx, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=(0.0, 1.0), topology=topo)
m1 = dist.LibraryDistributedModel(grid=g1, tracers=("T", "S"))
g2 = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=(0.0, 1.0), topology=topo)
m2 = ocn.NonhydrostaticModel(grid=g2, tracers=("T", "S"))
for m, g in ((m1, g1.local), (m2, g2)):
    vals = smooth_state({n: g.nodes(f.loc) for n, f in m.fields().items()}, seed=99)
    if offset is not None:
        vals["S"] = vals["S"] - 35.0 + offset
    ocn.set_model(m, **vals)
dt = 0.1 / size[1] / 0.6
for step in range(4):
    errs = {}
    for n in m1.fields():
        a, b = m1.fields()[n].parent()[3:-3, 3:-3, 3:-3], m2.fields()[n].parent()[3:-3, 3:-3, 3:-3]
        errs[n] = np.abs(a - b).max()
    print(step, {k: "%.2e" % v for k, v in errs.items()}, flush=True)
    ocn.time_step(m1, dt); ocn.time_step(m2, dt)
