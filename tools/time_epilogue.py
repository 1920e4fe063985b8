"""ms per time-step of the configs[4] physics workload (bench.py --workload ppb_amd: 256 x 256 x 128, AMD closure, buoyancy, Flux conditions) for
values of a library option (GPU box): python tools/time_epilogue.py epilogue_kchunk 8 16 32 64  |  python tools/time_epilogue.py epilogue_march 1 0"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oldoceananigans_jl_amd as ocn
from helpers import smooth_state, tanh_faces
import bench
opt, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
workload = os.environ.get("OCN_WORKLOAD", "ppb_amd")
arch = ocn.GPU(0)
N = 256
for rep in range(2):
    for val in vals:
        ocn.set_option(opt, val)
        grid = ocn.RectilinearGrid(arch, size=(N, N, N // 2), x=(0.0, 1.0), y=(0.0, 1.0), z=tanh_faces(N // 2), topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
        model = ocn.NonhydrostaticModel(grid=grid, advection=ocn.WENO(), tracers=("T", "S"), **bench.workload_physics(ocn, workload))
        ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in model.fields().items()}, 1234))
        dt = 0.1 / N / 0.6
        for _ in range(5): ocn.time_step(model, dt)
        ocn.synchronize(); t0 = time.perf_counter()
        for _ in range(30): ocn.time_step(model, dt)
        ocn.synchronize()
        print(f"{workload}: {opt} = {val}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
        model.close()
