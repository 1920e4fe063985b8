"""ms/step of the single-GPU model on an (Nx, Ny, Nz) grid, z Periodic or stretched Bounded (GPU box):
python tools/time_step_shape.py Nx Ny Nz [bounded] [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state, tanh_faces
shape = tuple(int(v) for v in sys.argv[1:4]); bounded = len(sys.argv) > 4 and sys.argv[4] == "bounded"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
arch = ocn.GPU(0)
for kv in filter(None, os.environ.get("OCN_SET_OPTIONS", "").split(",")):      # library options for A/B runs: OCN_SET_OPTIONS=key=value,...
    ocn.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if bounded:
    grid = ocn.RectilinearGrid(arch, size=shape, x=(0.0, 1.0), y=(0.0, 1.0), z=tanh_faces(shape[2]), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
else:
    grid = ocn.RectilinearGrid(arch, size=shape, extent=(1, 1, 1))
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 1234))
dt = 0.1 / max(shape) / 0.6
for _ in range(3): ocn.time_step(model, dt)
for rep in range(2):
    ocn.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): ocn.time_step(model, dt)
    ocn.synchronize()
    print(f"{shape} {'PPB' if bounded else 'PPP'}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step, max|div u| {ocn.max_abs_divergence(model):.2e}", flush=True)
