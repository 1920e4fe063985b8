"""diagnostic: one case of tests/test_gpu_parity.py::test_seeded_random_physics_configurations_match_the_oracle rebuilt (same seeded
generator), tendencies of HIP (epilogue_march 1 and 0) against the oracle: where they differ.  python tools/diag_random_case.py <case>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oldoceananigans_jl_amd as ocn
import oracle
from helpers import set_both, tanh_faces
want = int(sys.argv[1]) if len(sys.argv) > 1 else 0
arch = ocn.GPU(0)
rng = np.random.default_rng(771)
F = ocn.FieldBoundaryConditions
topo_names = ("Periodic", "Bounded")
sides = {0: ("west", "east"), 1: ("south", "north"), 2: ("bottom", "top")}
normal = {"u": 0, "v": 1, "w": 2}
for case in range(want + 1):
    size = tuple(int(rng.integers(4, 21)) for _ in range(3))
    topology = tuple(topo_names[int(rng.random() < 0.5)] for _ in range(3))
    z = tanh_faces(size[2]) if (topology[2] == "Bounded" and rng.random() < 0.5) else ((-1.0, 0.0) if topology[2] == "Bounded" else (0.0, 1.0))
    ntr = int(rng.integers(0, 3))
    gpu_names = ("T", "S")[:ntr]
    cpu_names = ["u", "v", "w"] + ["c%d" % t for t in range(ntr)]
    amd = rng.random() < 0.5
    kw = dict(closure=ocn.AnisotropicMinimumDissipation() if amd else ocn.ScalarDiffusivity(ν=3e-3, κ=2e-3))
    ops = [("amd",) if amd else ("closure",)]
    if rng.random() < 0.5:
        kw["coriolis"] = ocn.FPlane(f=0.6); ops.append(("coriolis",))
    if ntr == 2 and rng.random() < 0.6:
        kw["buoyancy"] = ocn.SeawaterBuoyancy(ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4)); ops.append(("seawater",))
    elif ntr >= 1 and rng.random() < 0.4:
        gpu_names = ("b",) + gpu_names[1:]
        kw["buoyancy"] = ocn.BuoyancyTracer(); ops.append(("btracer",))
    bcs = {}
    for name, cname in zip(("u", "v", "w") + gpu_names, cpu_names):
        conds = {}
        for d in range(3):
            if topology[d] == "Bounded" and normal.get(name) != d:
                for sd in sides[d]:
                    if rng.random() < 0.3:
                        val = float(rng.normal()) * 1e-3
                        conds[sd] = ocn.FluxBoundaryCondition(val)
                        ops.append(("bc", cname, sd, val))
        if conds:
            bcs[name] = F(**conds)
    if ntr >= 1 and topology[2] == "Bounded" and gpu_names[-1] not in bcs and rng.random() < 0.5:
        rate = 2.5e-3
        bcs[gpu_names[-1]] = F(top=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(b=-rate), field_dependencies=gpu_names[-1]))
        ops.append(("lin", cpu_names[-1], rate))
    if bcs:
        kw["boundary_conditions"] = bcs
print(case, size, topology, ntr, ops)
g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=z, topology=tuple(getattr(ocn, t) for t in topology))
g_cpu = oracle.Grid(size, topology=tuple({"Periodic": 0, "Bounded": 1}[t] for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=z)
m_cpu = oracle.Model(g_cpu, ntr)
for op in ops:
    if op[0] == "amd": m_cpu.set_amd()
    elif op[0] == "closure": m_cpu.set_closure(nu=3e-3, kappa=2e-3)
    elif op[0] == "coriolis": m_cpu.set_coriolis(0.6)
    elif op[0] == "seawater": m_cpu.set_seawater_buoyancy(alpha=2e-4, beta=8e-4)
    elif op[0] == "btracer": m_cpu.set_buoyancy_tracer(0)
    elif op[0] == "bc": m_cpu.set_bc(op[1], op[2], "flux", op[3])
    elif op[0] == "lin": m_cpu.set_linear_flux_bc(op[1], "top", 0.0, -op[2], op[1])
outs = {}
for march in (1, 0):
    ocn.set_option("epilogue_march", march); ocn.set_option("amd_march", march)
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, tracers=gpu_names, **kw)
    set_both(ocn, m_gpu, m_cpu, seed=900 + case, enforce_incompressibility=False)
    ocn.update_state(m_gpu, True)
    outs[march] = {n: m_gpu.tendency(n).parent() for n in m_gpu.fields()}
    names = list(m_gpu.fields().keys())
    m_gpu.close()
m_cpu.update_state(True)
for n, cn in zip(names, cpu_names):
    ref = m_cpu.field("G" + cn)
    for march in (1, 0):
        d = np.abs(outs[march][n] - ref)
        idx = np.argwhere(d > 0)
        print(f"G{n} march={march}: max diff {d.max():.3e} at {np.unravel_index(d.argmax(), d.shape)} (parent index), {len(idx)} cells differ; ref there {ref[np.unravel_index(d.argmax(), d.shape)]:.6e}")
        if len(idx):
            print("   i range", idx[:, 0].min(), idx[:, 0].max(), "j range", idx[:, 1].min(), idx[:, 1].max(), "k range", idx[:, 2].min(), idx[:, 2].max())
