"""one line: role tendency kernel (tendency_impl = 2) at N^3, plain launch, ms per evaluation (GPU box; used by tools/ab_run.sh)
python tools/time_roles.py [N] [kchunk]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shape = tuple(int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (N, N, N)
arch = ocn.GPU(0)
grid = ocn.RectilinearGrid(arch, size=shape, extent=(1, 1, 1))
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
flds = model.fields()
ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in flds.items()}, 1234))
model.set_option("tendency_impl", 2)
arith = int(os.environ.get("OCN_ARITHMETIC", "0"))      # 1: the opt-in contracted WENO flux
ocn.set_option("arithmetic", arith)
model.set_option("role_kchunk", kc)
for _ in range(5): ocn.update_state(model, True)
ocn.synchronize()
res = []
for rep in range(6):
    model.set_option("profile", 1)
    for _ in range(20): ocn.update_state(model, True)
    ms, n = model.profile_read(); model.set_option("profile", 0)
    res.append(ms / n)
dt = 0.1 / N / 0.6
for _ in range(2): ocn.time_step(model, dt)
avg = []
for rep in range(6):
    model.set_option("profile", 1)
    for _ in range(6): ocn.time_step(model, dt)
    ms, n = model.profile_read(); model.set_option("profile", 0)
    avg.append(ms / n)
print("arithmetic %d role kernel %s kchunk %d: plain min %.3f median %.3f | in time_step (2 of 3 with substep) min %.3f median %.3f ms" %
      (arith, "x".join(map(str, shape)), kc, min(res), sorted(res)[len(res) // 2], min(avg), sorted(avg)[len(avg) // 2]), flush=True)
