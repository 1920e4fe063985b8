"""run the tendency evaluation repeatedly (for rocprofv3): python tools/run_tendency.py N impl ty kchunk minw reps"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state
N, impl, ty, kc, mw, reps = (int(x) for x in sys.argv[1:7])
arch = ocn.GPU(0)
grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
flds = model.fields()
ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in flds.items()}, 1234))
for k, v in (("tendency_impl", impl), ("fused_ty", ty), ("fused_kchunk", kc), ("fused_minw", mw)):
    model.set_option(k, v)
model.set_option("profile", 1)
for _ in range(reps): ocn.update_state(model, True)
ms, n = model.profile_read()
print(f"{ms/n:.3f} ms/eval")
