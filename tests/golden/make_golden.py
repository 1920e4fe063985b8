#!/usr/bin/env python3
"""Generates the golden fixtures in this directory with the CPU oracle (oracle/).

The reference holds NO usable golden vectors for this path: its regression .jld2 files are remote DataDeps
(test/data_dependencies.jl:17-38), cover AB2 + Centered advection only, and Julia cannot run here. These fixtures therefore
freeze the ORACLE's output on small seeded cases so that (i) any later change of the oracle is caught on CPU and (ii) the
HIP path is checked against stored data in addition to the live oracle. Inputs are generated from the seed by
tests/helpers.py: only the expected outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from helpers import smooth_state, tanh_faces  # noqa: E402

CASES = {
    "ppp_16": dict(size=(16, 16, 16), topology=(0, 0, 0), z=(0.0, 1.0), nsteps=3),
    "ppb_stretched_16x16x12": dict(size=(16, 16, 12), topology=(0, 0, 1), z="tanh", nsteps=3),
    "bbb_12x10x8": dict(size=(12, 10, 8), topology=(1, 1, 1), z=(-1.0, 0.0), nsteps=3),      # cosine-transform pressure solve
}


def oracle_nodes(g, loc):
    out = []
    for d in range(3):
        n = g.N[d] + (1 if (loc[d] == 1 and g.topo[d] == 1) else 0)
        if d == 2 and not np.all(g.dc[2] == g.dc[2][0]):
            faces = np.concatenate([[0.0], np.cumsum(g.dc[2][3:3 + g.N[2]])]) - g.L[2]
            arr = faces[:n] if loc[d] == 1 else 0.5 * (faces[:-1] + faces[1:])
        else:
            origin = -g.L[2] if (d == 2 and g.topo[2] == 1) else 0.0
            arr = origin + g.dc[d][0] * (np.arange(n) + (0.0 if loc[d] == 1 else 0.5))
        shape = [1, 1, 1]
        shape[d] = n
        out.append(np.asarray(arr).reshape(shape))
    return out


def run_case(name):
    c = CASES[name]
    z = tanh_faces(c["size"][2]) if c["z"] == "tanh" else c["z"]
    if c["topology"][2] == 1 and c["z"] != "tanh":
        z = (-1.0, 0.0)
    g = O.Grid(c["size"], topology=c["topology"], z=z)
    m = O.Model(g, 2)
    locs = {"u": (1, 0, 0), "v": (0, 1, 0), "w": (0, 0, 1), "T": (0, 0, 0), "S": (0, 0, 0)}
    vals = smooth_state({k: oracle_nodes(g, l) for k, l in locs.items()}, seed=1234)
    m.set(u=vals["u"], v=vals["v"], w=vals["w"], c0=vals["T"], c1=vals["S"])
    dt = 0.1 * g.dc[0][0] / 0.6
    for _ in range(c["nsteps"]):
        m.time_step(dt)
    out = {k: m.field(n).copy() for k, n in (("u", "u"), ("v", "v"), ("w", "w"), ("T", "c0"), ("S", "c1"), ("p", "p"))}
    out["dt"] = dt
    return g, m, out


if __name__ == "__main__":
    for name in CASES:
        _, _, out = run_case(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
