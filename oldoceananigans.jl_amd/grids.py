"""RectilinearGrid (reference: src/Grids/rectilinear_grid.jl:3-25,264-291; coordinate generation
src/Grids/grid_generation.jl:34-135). Host-side metadata only; the device view lives behind `ocn_grid_t`."""
import ctypes as C
from fractions import Fraction

import numpy as np

from . import _lib


class Periodic:
    code = 0


class Bounded:
    code = 1


class FullyConnected:
    """a partitioned Periodic direction (distributed_grids.jl:339-346): halos are owned by the neighbouring ranks"""
    code = 2


class Flat:
    code = 3


class RightConnected:
    """first rank of a Bounded partitioned direction: a wall on the left, a neighbour on the right (distributed_grids.jl:339-346)"""
    code = 4


class LeftConnected:
    """last rank: a neighbour on the left, a wall on the right; Face fields hold N + 1 points like on Bounded (grid_utils.jl:43-68)"""
    code = 5


class Center:
    code = 0


class Face:
    code = 1


def _topo_code(t):
    t = t if isinstance(t, type) else type(t)
    return t.code


def _regular_coordinate(interval, N, name):
    """generate_coordinate(FT, topo, N, H, node_interval::Tuple) (grid_generation.jl:98-135): Δ = FT(BigFloat(L)/N)."""
    c1, c2 = Fraction(float(interval[0])), Fraction(float(interval[1]))
    if c2 < c1:
        raise ValueError(f"{name} must be an increasing interval!")
    L = c2 - c1
    return float(L / N), float(L), float(c1)


def _add12(x, y):
    """Base.add12 (twiceprecision.jl): x + y as an unevaluated sum hi + lo"""
    if abs(y) > abs(x):
        x, y = y, x
    hi = x + y
    return hi, y - (hi - x)


def _truncbits(x, nb):
    import struct
    b = struct.unpack("<Q", struct.pack("<d", x))[0] & ((0xFFFFFFFFFFFFFFFF << nb) & 0xFFFFFFFFFFFFFFFF)
    return struct.unpack("<d", struct.pack("<Q", b))[0]


def _rat(x):
    """Base.rat (twiceprecision.jl): a rational approximation n / d of a Float64 with |n|, |d| <= 2^24 by continued fractions; (n, 0)-like
    failures are reported by d == 0 or by n / d != x at the call site"""
    import math
    y = x
    a = d = 1
    b = c = 0
    m = 16777216.0                                   # maxintfloat(Float32)
    while abs(y) <= m:
        f = math.trunc(y)
        y -= f
        a, c = f * a + c, a
        b, d = f * b + d, b
        if max(abs(a), abs(b)) > 16777216:
            return c, d
        if b != 0 and a / b == x:
            break
        if y == 0:
            break
        y = 1.0 / y
    return a, b


def julia_range(start, stop, length):
    """`range(start, stop, length = length)` for Float64 end points as Julia evaluates it (base/twiceprecision.jl,
    `range_start_stop_length`): every element, rounded like `r[i]`. End points that are exact small rationals (0.25, -0.3, 1.2 ...) take
    the rational path -- the elements are the correctly rounded values of (start_n (len - i) + stop_n (i - 1)) / (den (len - 1)) --, others
    (multiples of π ...) the floating-point path `_linspace`: a StepRangeLen with twice-precision reference and step whose reference element
    is the one of smallest magnitude. The grids take their node coordinates from such ranges (grid_generation.jl:118-119), so a node is
    not `c₁ + (i - 1) Δ` in floating point -- e.g. the first x face of `RectilinearGrid(size=(32, 32), extent=(2π, 4π), ...)` is
    3.60072e-17 (rectilinear_grid.jl:194, reproduced by tests/test_reference_kats.py)."""
    import math
    if length == 1:
        return np.array([start], dtype=np.float64)
    sn, sd = _rat(start)
    en, ed = _rat(stop)
    if sd != 0 and ed != 0:
        den = sd * ed // math.gcd(sd, ed)
        if den != 0 and abs(den * start) <= 2.0 ** 53 and abs(den * stop) <= 2.0 ** 53:
            start_n, stop_n = int(round(den * start)), int(round(den * stop))
            if start_n / den == start and stop_n / den == stop:
                return np.array([float(Fraction(start_n * (length - i) + stop_n * (i - 1), den * (length - 1))) for i in range(1, length + 1)])
    d = stop - start
    imin = int(round(-(start / d) * (length - 1) + 1)) if d != 0 else 1        # round half to even, like round(Int, x)
    if 1 < imin < length:
        t = (imin - 1) / (length - 1)
        ref = (1 - t) * start + t * stop
        step = (ref - start) / (imin - 1) if imin - 1 < length - imin else (stop - ref) / (length - imin)
    elif imin <= 1:
        imin, ref, step = 1, start, d / (length - 1)
    else:
        imin, ref, step = length, stop, d / (length - 1)
    nb = min(27, math.ceil(math.log2(max(imin - 1, length - imin))))            # nbitslen(Float64, len, offset)
    step_hi = _truncbits(step, nb)
    x1_hi, x1_lo = _add12((1 - imin) * step_hi, ref)
    x2_hi, x2_lo = _add12((length - imin) * step_hi, ref)
    a, b = (start - x1_hi) - x1_lo, (stop - x2_hi) - x2_lo
    step_lo = (b - a) / (length - 1)
    ref_lo = a - (1 - imin) * step_lo
    out = np.empty(length, dtype=np.float64)
    for i in range(1, length + 1):
        u = i - imin
        x_hi, x_lo = _add12(ref, u * step_hi)
        out[i - 1] = x_hi + (x_lo + (u * step_lo + ref_lo))
    return out


def _regular_nodes(interval, N, H, topology):
    """the face and centre coordinates of generate_coordinate (grid_generation.jl:104-125), halos included: index H = node 1"""
    c1, c2 = Fraction(float(interval[0])), Fraction(float(interval[1]))
    L = c2 - c1
    D = L / N
    bounded = topology in (Bounded, LeftConnected)
    Fm = c1 - H * D
    Fp = Fm + (L + 2 * H * D if bounded else L + (2 * H - 1) * D)               # total_extent (grid_utils.jl:120-121)
    Cm = Fm + D / 2
    Cp = Cm + L + D * (2 * H - 1)
    TF, TC = N + 2 * H + (1 if bounded else 0), N + 2 * H                        # total_length (grid_utils.jl:63-65)
    return julia_range(float(Fm), float(Fp), TF), julia_range(float(Cm), float(Cp), TC)


def _stretched_coordinate(faces, N, H, bounded, name):
    """generate_coordinate for an explicit face vector / function (grid_generation.jl:34-95).
    Returns L, faces-with-halo, Δᶜ and Δᶠ as arrays indexed by position k-1+H for k = 1-H .. N+H+1 (entries the
    reference leaves undefined repeat the last defined value)."""
    if callable(faces):
        faces = [faces(k) for k in range(1, N + 2)]
    Fi = np.asarray(faces, dtype=np.float64)
    if Fi.shape != (N + 1,):
        raise ValueError(f"length({name}) must be N+1 = {N + 1}")
    if not np.all(np.diff(Fi) >= 0) or not np.all(np.diff(Fi) > 0):
        raise ValueError(f"The elements of {name} must be increasing!")
    L = Fi[N] - Fi[0]
    if bounded:
        dm = [Fi[1] - Fi[0] for _ in range(H)]
        dp = [Fi[N] - Fi[N - 1] for _ in range(H)]
    else:
        dm = [Fi[N - H + i] - Fi[N - H + i - 1] for i in range(1, H + 1)]
        dp = [Fi[i] - Fi[i - 1] for i in range(1, H + 1)]
    dp = dp[::-1]

    def lsum(v):
        s = v[0]
        for x in v[1:]:
            s = s + x
        return s

    Fm = [Fi[0] - lsum(dm[i:]) for i in range(H)]
    Fp = [Fi[N] + lsum(dp[i:]) for i in range(H)][::-1]
    F = np.concatenate([Fm, Fi, Fp])
    TC = N + 2 * H
    Cn = np.array([(F[i + 1] + F[i]) / 2 for i in range(TC)])
    dF = [Cn[i] - Cn[i - 1] for i in range(1, TC)]
    TF = N + 2 * H + (1 if bounded else 0)
    F = F[:TF]
    dC = [F[i + 1] - F[i] for i in range(TF - 1)]
    dF = [dF[0]] + dF + [dF[-1]]
    for i in range(len(dF) - 1, 0, -1):
        dF[i] = dF[i - 1]
    n = N + 2 * H + 1
    dzc = np.array([dC[min(p, len(dC) - 1)] for p in range(n)])                 # ref index starts at 1-H
    dzf = np.array([dF[min(p + 1, len(dF) - 1)] for p in range(n)])             # ref index starts at -H
    return float(L), F, Cn, dzc, dzf


class RectilinearGrid:
    """RectilinearGrid(arch; size, x, y, z | extent, topology, halo) -- x and y must be regular (2-tuples); z may be a
    2-tuple or an array/function of Nz+1 faces (Bounded only)."""

    def __init__(self, architecture, size, x=None, y=None, z=None, extent=None,
                 topology=(Periodic, Periodic, Periodic), halo=None):
        self.architecture = architecture
        # constructor_arguments(grid) (rectilinear_grid.jl:404-440): what with_halo / on_architecture rebuild the grid from
        self._constructor_arguments = dict(size=size, x=x, y=y, z=z, extent=extent, topology=topology)
        self.topology = tuple(t if isinstance(t, type) else type(t) for t in topology)
        flat = [t is Flat for t in self.topology]
        # Flat directions (Grids/input_validation.jl): `size`, `halo` and `extent` list the non-Flat directions only (full
        # 3-tuples with 1 / 0 entries are accepted too); a Flat direction has one cell, no halo, unit spacing
        def expand(values, fill, what):
            values = tuple(values) if isinstance(values, (tuple, list)) else (values,)
            if len(values) == 3:
                return tuple(fill if f else v for v, f in zip(values, flat))
            if len(values) != 3 - sum(flat):
                raise ValueError(f"{what} must have {3 - sum(flat)} (non-Flat directions) or 3 entries")
            it = iter(values)
            return tuple(fill if f else next(it) for f in flat)
        self.Nx, self.Ny, self.Nz = (int(n) for n in expand(size, 1, "size"))
        if halo is None:          # validate_halo(TX, TY, TZ, size, ::Nothing) (input_validation.jl:71-77): min(3, size)
            halo = tuple(0 if f else min(3, n) for n, f in zip(self.size, flat))
        self.Hx, self.Hy, self.Hz = (int(h) for h in expand(halo, 0, "halo"))
        for name, H, N in (("x", self.Hx, self.Nx), ("y", self.Hy, self.Ny)):      # input_validation.jl:86-92 (x and y only)
            if not H <= N:
                raise ValueError(f"halo={H} must be ≤ size={N} for coordinate {name}")
        if not self.Hz <= self.Nz:
            # the reference's validate_halo does not look at z; its halo fills then index outside the array. The library refuses such a
            # grid (ocn_grid_create) -- say so here, where the grid is written down, not when the lazy handle is first touched
            raise ValueError(f"halo={self.Hz} must be ≤ size={self.Nz} for coordinate z (the reference checks x and y only, input_validation.jl:86-92; "
                             "libocn_mi355x needs it in z too)")
        if extent is not None:
            if any(c is not None for c in (x, y, z)):
                raise ValueError("Cannot specify both extent and x, y, z keyword arguments!")
            # default_horizontal_extent / default_vertical_extent (Grids/input_validation.jl:161-162)
            ext = expand(extent, 1.0, "extent")
            x, y, z = (0.0, float(ext[0])), (0.0, float(ext[1])), (-float(ext[2]), 0.0)
        x, y, z = ((0.0, 1.0) if f else c for c, f in zip((x, y, z), flat))
        if x is None or y is None or z is None:
            raise ValueError("Must supply extent or x, y, z keyword when topology is not Flat")
        for name, c in (("x", x), ("y", y)):
            if not (isinstance(c, tuple) and len(c) == 2):
                raise NotImplementedError(f"stretched {name} is outside the accelerated hot path (only z may be stretched)")
        self.Δxᶜᵃᵃ, self.Lx, self.x0 = _regular_coordinate(x, self.Nx, "x")
        self.Δyᵃᶜᵃ, self.Ly, self.y0 = _regular_coordinate(y, self.Ny, "y")
        self.Δxᶠᵃᵃ, self.Δyᵃᶠᵃ = self.Δxᶜᵃᵃ, self.Δyᵃᶜᵃ
        # node coordinates with halos (grid.xᶠᵃᵃ, xᶜᵃᵃ, yᵃᶠᵃ, yᵃᶜᵃ; array position H holds node 1): Julia ranges, see julia_range
        # (the USER's end points go in: c₁, c₂ = BigFloat.(node_interval), grid_generation.jl:104-105 -- x0 + Lx is not always c₂ in Float64)
        self.xᶠᵃᵃ, self.xᶜᵃᵃ = (np.full(1, self.x0),) * 2 if flat[0] else _regular_nodes((float(x[0]), float(x[1])), self.Nx, self.Hx, self.topology[0])
        self.yᵃᶠᵃ, self.yᵃᶜᵃ = (np.full(1, self.y0),) * 2 if flat[1] else _regular_nodes((float(y[0]), float(y[1])), self.Ny, self.Hy, self.topology[1])
        n = self.Nz + 2 * self.Hz + 1
        if isinstance(z, tuple) and len(z) == 2 and np.isscalar(z[0]):
            dz, self.Lz, self.z0 = _regular_coordinate(z, self.Nz, "z")
            self.z_regular = True
            self.Δzᵃᵃᶜ = np.full(n, dz)
            self.Δzᵃᵃᶠ = np.full(n, dz)
            self._dz = dz
            self.zᵃᵃᶠ, self.zᵃᵃᶜ = (np.full(1, self.z0),) * 2 if flat[2] else _regular_nodes((float(z[0]), float(z[1])), self.Nz, self.Hz, self.topology[2])
        else:
            if self.topology[2] is not Bounded:
                raise NotImplementedError("a stretched z coordinate requires a Bounded z topology")
            self.Lz, self.zᵃᵃᶠ, self.zᵃᵃᶜ, self.Δzᵃᵃᶜ, self.Δzᵃᵃᶠ = _stretched_coordinate(
                z, self.Nz, self.Hz, True, "z")
            self.z_regular = False
            self._dz = 0.0
        self._handle = None

    @property
    def handle(self):
        """the library's device view of the grid, created on first use: a grid whose halo is smaller than the advection scheme
        needs may exist as host metadata (the model constructor replaces it, `inflate_grid_halo_size`), but the library refuses it"""
        if self._handle is None:
            h = C.c_void_p()
            dp = C.POINTER(C.c_double)
            zc = None if self.z_regular else self.Δzᵃᵃᶜ.ctypes.data_as(dp)
            zf = None if self.z_regular else self.Δzᵃᵃᶠ.ctypes.data_as(dp)
            _lib.check(_lib.lib().ocn_grid_create(
                C.byref(h), _lib.i3(self.size), _lib.i3(self.halo_size), _lib.i3([_topo_code(t) for t in self.topology]),
                (C.c_double * 3)(self.Lx, self.Ly, self.Lz), self.Δxᶜᵃᵃ, self.Δyᵃᶜᵃ, self._dz, zc, zf))
            self._handle = h
        return self._handle

    @property
    def size(self):
        return (self.Nx, self.Ny, self.Nz)

    @property
    def halo_size(self):
        return (self.Hx, self.Hy, self.Hz)

    def total_size(self, loc):
        """total_size(loc, topo, N, H) (grid_utils.jl:138-169)"""
        return tuple(n + 2 * h + (1 if (l is Face and t in (Bounded, LeftConnected)) else 0)
                     for n, h, l, t in zip(self.size, self.halo_size, loc, self.topology))

    def interior_size(self, loc):
        return tuple(n + (1 if (l is Face and t in (Bounded, LeftConnected)) else 0) for n, l, t in zip(self.size, loc, self.topology))

    def nodes(self, loc):
        """interior node coordinates (xnodes, ynodes, znodes) as broadcastable arrays"""
        out = []
        for d, (n, l, t) in enumerate(zip(self.size, loc, self.topology)):
            delta = (self.Δxᶜᵃᵃ, self.Δyᵃᶜᵃ, self._dz)[d]
            origin = (self.x0, self.y0, getattr(self, "z0", 0.0))[d]
            m = n + (1 if (l is Face and t in (Bounded, LeftConnected)) else 0)
            if d == 2 and not self.z_regular:
                arr = self.zᵃᵃᶠ[self.Hz:self.Hz + m] if l is Face else self.zᵃᵃᶜ[self.Hz:self.Hz + m]
            elif t is Flat:
                arr = np.full(m, origin)
            else:
                H = self.halo_size[d]
                F, Cn = ((self.xᶠᵃᵃ, self.xᶜᵃᵃ), (self.yᵃᶠᵃ, self.yᵃᶜᵃ), (self.zᵃᵃᶠ, self.zᵃᵃᶜ))[d]
                arr = (F if l is Face else Cn)[H:H + m]
            shape = [1, 1, 1]
            shape[d] = m
            out.append(np.asarray(arr, dtype=np.float64).reshape(shape))
        return out

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().ocn_grid_destroy(self._handle)
        except Exception:
            pass

    def __repr__(self):
        names = "×".join(str(n) for n in self.size)
        return f"{names} RectilinearGrid{{Float64, {', '.join(t.__name__ for t in self.topology)}}} on {self.architecture} with {self.halo_size} halo"


def with_halo(halo, grid):
    """with_halo(halo, grid::RectilinearGrid) (rectilinear_grid.jl:442-449): the same grid with another halo"""
    return type(grid)(grid.architecture, halo=tuple(halo), **grid._constructor_arguments)
