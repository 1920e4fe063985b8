"""Shared helpers for parity tests: seeded synthetic states set identically on the oracle and on the GPU model."""
import numpy as np


def tanh_faces(Nz, S=1.3, Lz=1.0):
    """hyperbolically spaced faces on [-Lz, 0] (form of test/test_time_stepping.jl:439-440)"""
    k = np.arange(Nz + 1)
    return -Lz + Lz * (np.tanh(S * (2 * k / Nz - 1)) / np.tanh(S) + 1) / 2


def random_state(shapes, seed=1234, amp=1.0):
    rng = np.random.default_rng(seed)
    return {name: amp * rng.standard_normal(shape) for name, shape in shapes.items()}


def smooth_state(grid_nodes, seed=1234):
    """Taylor-Green-like velocities + noise, Gaussian T, sinusoidal S"""
    rng = np.random.default_rng(seed)
    out = {}
    for name, (x, y, z) in grid_nodes.items():
        shape = np.broadcast(x, y, z).shape
        noise = 0.01 * rng.uniform(-1, 1, shape)
        if name == "u":
            out[name] = 0.5 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) * np.cos(2 * np.pi * z) + noise
        elif name == "v":
            out[name] = -0.5 * np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y) * np.cos(2 * np.pi * z) + noise
        elif name == "w":
            out[name] = 0.1 * np.cos(2 * np.pi * x) * np.cos(2 * np.pi * y) * np.sin(2 * np.pi * z) + noise
        elif name == "T":
            out[name] = np.exp(-((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - z.mean()) ** 2) / 0.02) + 1e-3 * noise
        else:
            # the z-dependence keeps the state generic: a tracer that is EXACTLY uniform along a direction (with a large offset)
            # makes the WENO smoothness indicators in that direction pure round-off, and the oracle itself then moves by
            # 1e-13 relative under sub-ulp perturbations of u (measured) -- conditioning of the scheme, not of an implementation
            # (hence one term per direction: on grids with Flat directions the product term alone can vanish)
            out[name] = (35 + np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) + 0.2 * np.cos(2 * np.pi * z) +
                         0.3 * np.sin(2 * np.pi * y) + 0.25 * np.cos(2 * np.pi * x))
    return out


def rel_err(a, b):
    """max |a-b| / max |b| (the 1e-12 relative tolerance of BASELINE.json's north_star is on this quantity)"""
    scale = np.max(np.abs(b))
    return np.max(np.abs(a - b)) / (scale if scale > 0 else 1.0)


ORACLE_TOPO = {"Periodic": 0, "Bounded": 1, "Flat": 3}


def make_pair(ocn, O, arch, size, topology=("Periodic", "Periodic", "Periodic"), z=None, ntracers=2):
    """build the same grid + model on the GPU (product) and on the CPU oracle"""
    topo_cls = tuple(getattr(ocn, t) for t in topology)
    zc = z if z is not None else ((-1.0, 0.0) if topology[2] == "Bounded" else (0.0, 1.0))
    g_gpu = ocn.RectilinearGrid(arch, size=size, x=(0.0, 1.0), y=(0.0, 1.0), z=zc, topology=topo_cls)
    g_cpu = O.Grid(size, topology=tuple(ORACLE_TOPO[t] for t in topology), x=(0.0, 1.0), y=(0.0, 1.0), z=zc)
    names = ("T", "S", "C3", "C4", "C5", "C6", "C7", "C8")[:ntracers]
    m_gpu = ocn.NonhydrostaticModel(grid=g_gpu, advection=ocn.WENO(), tracers=names)
    m_cpu = O.Model(g_cpu, ntracers)
    return g_gpu, g_cpu, m_gpu, m_cpu


def set_both(ocn, m_gpu, m_cpu, seed=1234, smooth=False, enforce_incompressibility=True):
    g = m_gpu.grid
    flds = m_gpu.fields()
    if smooth:
        nodes = {n: g.nodes(f.loc) for n, f in flds.items()}
        vals = smooth_state(nodes, seed)
    else:
        vals = random_state({n: g.interior_size(f.loc) for n, f in flds.items()}, seed)
    ocn.set_model(m_gpu, enforce_incompressibility=enforce_incompressibility, **vals)
    cpu_names = ["u", "v", "w"] + ["c%d" % t for t in range(len(m_gpu.tracer_names))]
    m_cpu.set(enforce_incompressibility=enforce_incompressibility, **{cn: vals[gn] for cn, gn in zip(cpu_names, flds.keys())})
    return vals


def field_pairs(m_gpu, m_cpu):
    """yield (name, gpu parent array, oracle parent array)"""
    names = list(m_gpu.fields().keys())
    cpu_names = ["u", "v", "w"] + ["c%d" % t for t in range(len(m_gpu.tracer_names))]
    for gn, cn in zip(names, cpu_names):
        yield gn, m_gpu.fields()[gn].parent(), m_cpu.field(cn)
    yield "pNHS", m_gpu.pressures.pNHS.parent(), m_cpu.field("p")
