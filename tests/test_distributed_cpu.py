"""N > 1 path on CPU: world_size-2 gloo runs of the PRODUCT's distributed orchestration (oldoceananigans.jl_amd.distributed)
on the test-only CPU backend, checked against the serial oracle on the global grid.

Mirrors the reference's distributed tests: rank connectivity (test_distributed_models.jl:110-223), local grid
extents (:225-280), halo exchange with rank ids and exact `==` (:334-404), transpose round trip (test_distributed_transpose.jl:
13-54) and distributed-vs-serial agreement of the model (the reference compares with `≈`)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_workers(tmp_path, nsteps, async_halos, size, port):
    env = dict(os.environ, OMP_NUM_THREADS="2", OCN_ORACLE_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(nsteps),
           str(int(async_halos))] + [str(s) for s in size]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2)]


def _serial(oracle, size, nsteps):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import analytic
    g = oracle.Grid(size, x=(0.0, 2.0), y=(0.0, 1.0), z=(0.0, 1.0))
    m = oracle.Model(g, 2)
    d = [2.0 / size[0], 1.0 / size[1], 1.0 / size[2]]
    vals = {}
    for name, cn, loc in (("u", "u", (1, 0, 0)), ("v", "v", (0, 1, 0)), ("w", "w", (0, 0, 1)), ("T", "c0", (0, 0, 0)),
                          ("S", "c1", (0, 0, 0))):
        ax = []
        for dim in range(3):
            shape = [1, 1, 1]
            shape[dim] = size[dim]
            ax.append((d[dim] * (np.arange(size[dim]) + (0.0 if loc[dim] else 0.5))).reshape(shape))
        vals[cn] = analytic(name, *ax)
    m.set(**vals)
    dt = 0.1 * d[0] / 0.6
    for _ in range(nsteps):
        m.time_step(dt)
    return m


@pytest.mark.parametrize("async_halos", [False, True, 5])
def test_two_rank_model_matches_serial_oracle(oracle, tmp_path, async_halos):
    """async_halos = 5: asynchronous exchange with 5-wide (wider than Hx) buffer strips, local Nx = 12"""
    size, nsteps = ((24, 8, 8) if async_halos == 5 else (16, 8, 8)), 3
    ranks = _run_workers(tmp_path, nsteps, async_halos, size, 29533 + int(async_halos))
    m = _serial(oracle, size, nsteps)
    nxl = size[0] // 2
    for r, data in enumerate(ranks):
        assert int(data["iteration"]) == nsteps and float(data["time"]) == m.time
        assert float(data["div"]) < 5e-8 and bool(data["gather_ok"]) and bool(data["ids_ok"])
        for name, cn in (("u", "u"), ("v", "v"), ("w", "w"), ("T", "c0"), ("S", "c1"), ("p", "p")):
            glob = m.field(cn)
            mine = data[name]
            ref = glob[3 + r * nxl:3 + (r + 1) * nxl, 3:-3, 3:-3]
            got = mine[3:-3, 3:-3, 3:-3]
            scale = np.abs(glob).max()
            assert np.abs(got - ref).max() <= 1e-12 * scale, (r, name, np.abs(got - ref).max() / scale)
            if name != "p":
                # x halos must hold the neighbour's interior (periodic ring): exact copies
                west = glob[3 + ((r * nxl - 3) % size[0]):, 3:-3, 3:-3][:3] if r == 0 else glob[3 + r * nxl - 3:3 + r * nxl, 3:-3, 3:-3]
                assert np.abs(mine[:3, 3:-3, 3:-3] - west).max() <= 1e-12 * scale


def test_partition_connectivity_and_extents():
    """rank connectivity with periodic wrap (distributed_architectures.jl:391-434) and local extents N / R"""
    import oldoceananigans_jl_amd as ocn  # noqa: F401  (imports the product package without touching the GPU)
    from oldoceananigans_jl_amd.distributed import Partition, partition_coordinate
    p = Partition(4)
    assert [p.neighbours(r) for r in range(4)] == [(3, 1), (0, 2), (1, 3), (2, 0)]
    assert Partition(2).neighbours(0) == (1, 1)
    assert Partition(4).neighbours(0, periodic=False) == (None, 1)
    edges = [partition_coordinate((0.0, 1.0), 4, 4, r) for r in range(4)]
    assert edges[0][0] == 0.0 and edges[3][1] == 1.0
    for a, b in zip(edges[:-1], edges[1:]):
        assert a[1] == b[0]
