"""time the interior / strip split of the tendency evaluation (distributed update_state) for different strip widths (GPU box)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oldoceananigans_jl_amd as ocn
from oldoceananigans_jl_amd import kernels
from helpers import smooth_state
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
arch = ocn.GPU(0)
grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"))
flds = model.fields()
ocn.set_model(model, **smooth_state({n: grid.nodes(f.loc) for n, f in flds.items()}, 1234))
F = list(flds.values())
G = [model.tendency(n) for n in flds]


def run(ranges, reps=10):
    for r in ranges:
        kernels.compute_tendencies(grid, F[0], F[1], F[2], F[3:], G[0], G[1], G[2], G[3:], r)
    ocn.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for r in ranges:
            kernels.compute_tendencies(grid, F[0], F[1], F[2], F[3:], G[0], G[1], G[2], G[3:], r)
    ocn.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


print(f"whole                : {run([None]):.3f} ms", flush=True)
for W in (3, 8, 32, 64):
    full = [(W + 1, N - W, 1, N, 1, N), (1, W, 1, N, 1, N), (N - W + 1, N, 1, N, 1, N)]
    print(f"strip width {W:3d}: interior {run(full[:1]):.3f}  west {run(full[1:2]):.3f}  east {run(full[2:]):.3f}  all three {run(full):.3f} ms", flush=True)
cols = [(1 + 64 * c, 64 * (c + 1), 1, N, 1, N) for c in range(N // 64)]
print(f"{len(cols)} x-columns of 64   : {run(cols):.3f} ms", flush=True)
halves = [(1, N // 2, 1, N, 1, N), (N // 2 + 1, N, 1, N, 1, N)]
print(f"2 x-halves          : {run(halves):.3f} ms", flush=True)
yh = [(1, N, 1, 126, 1, N), (1, N, 127, N, 1, N)]
print(f"2 y-halves (126+130): {run(yh):.3f} ms", flush=True)
zh = [(1, N, 1, N, 1, N // 2), (1, N, 1, N, N // 2 + 1, N)]
print(f"2 z-halves          : {run(zh):.3f} ms", flush=True)
for kc in (16, 22, 26, 32, 43, 52, 64):
    ocn.set_option("fused_kchunk", kc)
    print(f"kchunk {kc}: whole {run([None]):.3f}  4 columns {run(cols):.3f} ms", flush=True)
ocn.set_option("fused_kchunk", 0)
for _ in range(3):
    print(f"auto: whole {run([None]):.3f}  strips64 {run([(65, N - 64, 1, N, 1, N), (1, 64, 1, N, 1, N), (N - 63, N, 1, N, 1, N)]):.3f} ms", flush=True)
