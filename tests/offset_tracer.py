"""The survey's OWN synthetic inputs and the bound that holds for them.

SURVEY.md 8(d): u, v, w = Taylor-Green-like 3-D field + 0.01 U(-1, 1) noise; T = Gaussian blob + 1e-3 noise; S = 35 + sin(2 pi x) cos(2 pi y).
The other parity tests use `helpers.smooth_state`, whose S carries three more terms; this module keeps S exactly as the survey wrote it,
because that state shows a property of the REFERENCE's WENO formulation that no implementation can remove:

  the smoothness indicators are evaluated as sums of PRODUCTS of the stencil values (weno_interpolants.jl:204-216: beta =
  p0 (C1 p0 + C2 p1 + C3 p2) + p1 (C4 p1 + C5 p2) + C6 p2^2), not of differences. For a tracer with a large offset S0 every product is
  ~ S0^2 and beta is what is left after they cancel: its round-off is eps * O(50) * S0^2 ~ 1e-11 for S0 = 35 -- absolute, whatever the
  true beta is. S = 35 + sin(2 pi x) cos(2 pi y) is EXACTLY uniform along z everywhere, along x on the planes cos(2 pi y) = 0 and along y
  on the planes sin(2 pi x) = 0: there beta is nothing but that round-off, the nonlinear weights alpha = C (1 + (tau / (beta + 1e-8))^2)
  turn it into an O(1e-6) relative scatter of the weights, and a perturbation of the advecting velocity in its LAST BIT moves S by 1e-12
  .. 1e-11 of its value after one step (grows with the number of points per direction and with S0^2) -- 10^4 times what the same
  perturbation does to T or to the velocities. Measured on the oracle itself: tests/test_offset_tracer_sensitivity.py.

Consequence: north_star's "1e-12 relative" holds for u, v, w, T (and for S once it varies generically, helpers.smooth_state) but NOT for
the survey's S on directions of N >~ 64 points -- in ANY implementation, the reference's included: two exact-arithmetic-equivalent
evaluations (rocFFT vs the oracle's FFT, substructured vs single-GPU pressure solve) differ by `offset_tracer_bound`, not by 1e-12."""
import numpy as np


def survey_state(grid_nodes, seed=1234, offset=35.0):
    """the inputs of SURVEY.md 8(d) config 2, S = offset + sin(2 pi x) cos(2 pi y) UNCHANGED"""
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
            out[name] = offset + np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) + 0.0 * z
    return out


def offset_tracer_bound(size, nsteps, offset=35.0):
    """relative bound (max |dS| / max |S|) on the difference in S between two round-off-equivalent evaluations of `nsteps` RK3 steps from
    `survey_state` on a grid whose longest direction has N = max(size) points, at dt = 0.1 dx / max|u|:

        4e-12 * (N / 32)^2 * nsteps * (offset / 35)

    EMPIRICAL: fitted with a margin of >= 2 to the oracle's own sensitivity to a last-bit perturbation of the initial velocities
    (8.6e-13 at 32^3, 2.6e-12 at 64^3, 2.9e-12 .. 4.5e-12 on 32 x 128 x 128, 1.4e-12 .. 5.1e-12 on 16 x 256 x 256 over 1 .. 3 steps:
    tests/test_offset_tracer_sensitivity.py) and to the difference between the partitioned and the single-GPU pressure solver on the
    64 x 512 x 512 slab of configs[3] (5e-11 after one step, 7e-10 after three: tools/diag_slab.py, round 2). The relative error grows
    in proportion to the offset (absolute: to its square -- beta's round-off is eps * S0^2)."""
    N = max(size)
    return 4e-12 * (N / 32.0) ** 2 * nsteps * (abs(offset) / 35.0)


def cell_nodes(size, extent=(1.0, 1.0, 1.0)):
    """nodes of u, v, w, T, S on a triply periodic regular grid (oracle-side twin of RectilinearGrid.nodes)"""
    locs = {"u": (1, 0, 0), "v": (0, 1, 0), "w": (0, 0, 1), "T": (0, 0, 0), "S": (0, 0, 0)}
    out = {}
    for k, loc in locs.items():
        ax = []
        for d in range(3):
            shape = [1, 1, 1]
            shape[d] = size[d]
            ax.append((extent[d] * (np.arange(size[d]) + (0.0 if loc[d] else 0.5)) / size[d]).reshape(shape))
        out[k] = ax
    return out
