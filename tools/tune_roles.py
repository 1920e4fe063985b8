"""time the role-decomposed tendency kernel (tendency_impl = 2) against the all-fields kernel (impl 1) at 256^3 (GPU box):
python tools/tune_roles.py [N]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
arch = ocn.GPU(0)
grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
flds = model.fields()
ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in flds.items()}, 1234))
ref = None
variants = [(1, 7, 0, 0), (2, 1, 0, 0), (2, 1, 16, 0), (2, 1, 32, 0)]
for impl, ty, kc, dbg in variants:
    model.set_option("tendency_impl", impl)
    if impl == 2:
        model.set_option("role_kchunk", kc)
    for n in flds:
        model.tendency(n).set_parent(np.zeros(model.tendency(n).shape))
    ocn.update_state(model, True); ocn.synchronize()
    G = [model.tendency(n).parent() for n in flds]
    if ref is None: ref = G
    same = all(np.array_equal(a, b) for a, b in zip(G, ref))
    model.set_option("profile", 1)
    for _ in range(10): ocn.update_state(model, True)
    ms, n = model.profile_read(); model.set_option("profile", 0)
    print(f"impl {impl} rows {ty} kchunk {kc}: {ms/n:.3f} ms/eval  bit-identical-to-impl-1 {same}  -> {80*N**3/(ms/n*1e-3)/1e9:.0f} GB/s (80 B/cell)", flush=True)
model.set_option('role_kchunk', 0)
