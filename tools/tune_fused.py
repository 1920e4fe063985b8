"""time the tendency evaluation at 256^3 for different kernel variants (GPU box)"""
import sys, os, time
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
vals = smooth_state({n: grid.nodes(f.loc) for n, f in flds.items()}, 1234)
ocn.set_model(model, **vals)
ref = None
variants = [(0, 8, 16, 4, 0, 0), (1, 7, 0, 2, 1, 0), (1, 7, 0, 2, 1, 1), (1, 7, 32, 2, 1, 1), (1, 7, 52, 2, 1, 1), (1, 7, 22, 2, 1, 1)]
for impl, ty, kc, mw, zw, lds in variants:
    model.set_option("tendency_impl", impl); model.set_option("fused_ty", ty); model.set_option("fused_kchunk", kc); model.set_option("fused_minw", mw); model.set_option("fused_zwin", zw);
    ocn.update_state(model, True); ocn.synchronize()
    G = [model.tendency(n).parent() for n in flds]
    if ref is None: ref = G
    same = all(np.array_equal(a, b) for a, b in zip(G, ref))
    model.set_option("profile", 1)
    for _ in range(10): ocn.update_state(model, True)
    ms, n = model.profile_read(); model.set_option("profile", 0)
    print(f"impl {impl} ty {ty} kchunk {kc} minw {mw} zwin {zw} lds {lds}: {ms/n:.3f} ms/eval  bit-identical-to-v1 {same}  -> {80*N**3/(ms/n*1e-3)/1e9:.0f} GB/s algorithmic", flush=True)
