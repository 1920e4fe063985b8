"""diagnostic: per-step relative errors GPU vs oracle for one topology (usage: python tools/diag_parity.py B P B stretched)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import oldoceananigans_jl_amd as ocn  # noqa: E402
from oracle import oracle as O  # noqa: E402
from helpers import field_pairs, make_pair, rel_err, set_both, tanh_faces  # noqa: E402

names = {"P": "Periodic", "B": "Bounded", "F": "Flat"}
topo = tuple(names[a] for a in sys.argv[1:4])
stretched = len(sys.argv) > 4 and sys.argv[4] == "stretched"
size = tuple(1 if t == "Flat" else n for t, n in zip(topo, (12, 12, 10)))
arch = ocn.GPU(0)
z = tanh_faces(size[2]) if stretched else None
g_gpu, g_cpu, m_gpu, m_cpu = make_pair(ocn, O, arch, size, topo, z=z)
set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=True)
dt = 0.1 * min(d for d, t in zip((g_gpu.Δxᶜᵃᵃ, g_gpu.Δyᵃᶜᵃ, 0.1), topo) if t != 'Flat') / 0.6
core = tuple(slice(0, None) if t == "Flat" else slice(3, -3) for t in topo)
print("step 0", {n: f"{rel_err(a[core], b[core]):.2e}" for n, a, b in field_pairs(m_gpu, m_cpu)})
for s in range(10):
    ocn.time_step(m_gpu, dt)
    m_cpu.time_step(dt)
    print("step", s + 1, {n: f"{rel_err(a[core], b[core]):.2e}" for n, a, b in field_pairs(m_gpu, m_cpu)})
