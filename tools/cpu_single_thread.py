"""one RK3 step of the CPU oracle at 128^3 / 256^3 on ONE thread (context for the reference's published single-core number,
docs/src/appendix/benchmarks.md: 19.56 s per step at 256^3 for the older WENO benchmark) -- run on the GPU box's host"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from helpers import smooth_state
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
O.lib().oro_set_num_threads(1)
g = O.Grid((N, N, N))
m = O.Model(g, 2)
locs = {"u": (1, 0, 0), "v": (0, 1, 0), "w": (0, 0, 1), "T": (0, 0, 0), "S": (0, 0, 0)}
nodes = {k: [((np.arange(N) + (0.0 if l[d] else 0.5)) / N).reshape([N if e == d else 1 for e in range(3)]) for d in range(3)] for k, l in locs.items()}
v = smooth_state(nodes, 1234)
m.set(u=v["u"], v=v["v"], w=v["w"], c0=v["T"], c1=v["S"])
dt = 0.1 / N / 0.6
m.time_step(dt)
t0 = time.perf_counter()
m.time_step(dt)
el = time.perf_counter() - t0
print(f"oracle, 1 thread, {N}^3: {el:.2f} s per RK3 step = {N**3/el:.3e} cell-updates/s")
