"""Advection scheme descriptors and their adaptation to the grid (reference: src/Advection/centered_reconstruction.jl:5-21,
upwind_biased_reconstruction.jl:5-33, weno_reconstruction.jl:77-93, flux_form_advection.jl:4-42, adapt_advection_order.jl:18-96,
Advection.jl:63-65). Host-side metadata: the library evaluates WENO(order=5) and what adapt_advection_order turns it into."""


class Centered:
    """Centered(order=2): Centered{N} with N = order / 2 and buffer_scheme Centered(order - 2) (nothing for N = 1)."""

    def __init__(self, order=2):
        if order % 2 != 0:
            raise ValueError("Centered reconstruction scheme is defined only for even orders")
        self.order, self.buffer = order, order // 2
        self.buffer_scheme = Centered(order - 2) if self.buffer > 1 else None

    def summary(self):
        return f"Centered(order={2 * self.buffer})"

    def __eq__(self, other):
        return type(other) is Centered and other.buffer == self.buffer

    __hash__ = None

    def __repr__(self):
        return self.summary() + " \n└── buffer_scheme: " + (self.buffer_scheme.summary() if self.buffer_scheme else "Nothing")


class UpwindBiased:
    """UpwindBiased(order=3): UpwindBiased{N}, N = (order + 1) / 2; advecting_velocity_scheme Centered(order - 1) and buffer_scheme
    UpwindBiased(order - 2) for N > 1, Centered(order = 2) and nothing for N = 1."""

    def __init__(self, order=3):
        if order % 2 == 0:
            raise ValueError("UpwindBiased reconstruction scheme is defined only for odd orders")
        self.order, self.buffer = order, (order + 1) // 2
        if self.buffer > 1:
            self.advecting_velocity_scheme = Centered(order - 1)
            self.buffer_scheme = UpwindBiased(order - 2)
        else:
            self.advecting_velocity_scheme = Centered(2)
            self.buffer_scheme = None

    def summary(self):
        return f"UpwindBiased(order={2 * self.buffer - 1})"

    def __eq__(self, other):
        return type(other) is UpwindBiased and other.buffer == self.buffer

    __hash__ = None

    def __repr__(self):
        return (self.summary() + " \n├── buffer_scheme: " + (self.buffer_scheme.summary() if self.buffer_scheme else "Nothing") +
                "\n└── advecting_velocity_scheme: " + self.advecting_velocity_scheme.summary())


class WENO:
    """WENO(order=5): WENO{N, Float64, Float32}, N = (order + 1) / 2, buffer_scheme WENO(order - 2), advecting_velocity_scheme
    Centered(order - 1); WENO(order=1) IS UpwindBiased(order=1) (weno_reconstruction.jl:81-83). The library evaluates
    WENO(order=5) without bounds and the lower orders adapt_advection_order derives from it."""

    def __new__(cls, order=5, bounds=None):
        if order % 2 == 0:
            raise ValueError("WENO reconstruction scheme is defined only for odd orders")
        if order < 3:
            return UpwindBiased(order=1)
        return super().__new__(cls)

    def __init__(self, order=5, bounds=None):
        self.order, self.buffer, self.bounds = order, (order + 1) // 2, bounds
        self.advecting_velocity_scheme = Centered(order - 1)
        self.buffer_scheme = WENO(order=order - 2, bounds=bounds)

    def summary(self):
        return f"WENO{{{self.buffer}, Float64, Float32}}(order={2 * self.buffer - 1})"

    def __eq__(self, other):
        return type(other) is WENO and other.buffer == self.buffer and other.bounds == self.bounds

    __hash__ = None

    def __repr__(self):
        s = self.summary() + "\n"
        if self.bounds is not None:
            s += f"├── bounds: {self.bounds}\n"
        return (s + "├── buffer_scheme: " + self.buffer_scheme.summary() +
                "\n└── advection_velocity_scheme: " + self.advecting_velocity_scheme.summary())


class FluxFormAdvection:
    """FluxFormAdvection(x, y, z): one reconstruction scheme per direction (flux_form_advection.jl:4-27)"""

    def __init__(self, x, y, z):
        self.x, self.y, self.z = x, y, z
        self.buffer = max(required_halo_size_x(x), required_halo_size_y(y), required_halo_size_z(z))

    def summary(self):
        return f"FluxFormAdvection(x={self.x.summary()}, y={self.y.summary()}, z={self.z.summary()})"

    def __repr__(self):
        return ("FluxFormAdvection with direction-based reconstructions: \n    ├── x: " + self.x.summary() + "\n    ├── y: " +
                self.y.summary() + "\n    └── z: " + self.z.summary())


def _required(scheme, direction):
    if scheme is None:
        return 0
    if isinstance(scheme, FluxFormAdvection):                 # flux_form_advection.jl:40-42
        return getattr(scheme, direction).buffer
    return getattr(scheme, "buffer", 1)                       # required_halo_size_*(::AbstractAdvectionScheme{B}) = B; default 1


def required_halo_size_x(scheme):
    return _required(scheme, "x")


def required_halo_size_y(scheme):
    return _required(scheme, "y")


def required_halo_size_z(scheme):
    return _required(scheme, "z")


def _adapt_one(scheme, N, flat):
    """adapt_advection_order(topo, advection, N, grid) (adapt_advection_order.jl:62-96)"""
    if flat or scheme is None or not isinstance(scheme, (Centered, UpwindBiased, WENO)):
        return scheme
    if N >= scheme.buffer:
        return scheme
    if isinstance(scheme, Centered):
        return Centered(order=2 * N)
    if isinstance(scheme, UpwindBiased):
        return UpwindBiased(order=2 * N - 1)
    return WENO(order=2 * N - 1)


def adapt_advection_order(advection, grid):
    """adapt_advection_order(advection, grid) (adapt_advection_order.jl:18-50): the scheme itself when no direction changes,
    a FluxFormAdvection of the per-direction schemes otherwise."""
    if advection is None:
        return None
    from .grids import Flat
    parts = [getattr(advection, d) if isinstance(advection, FluxFormAdvection) else advection for d in "xyz"]
    new = [_adapt_one(s, N, t is Flat) for s, N, t in zip(parts, grid.size, grid.topology)]
    changed = any(not (a == b) for a, b in zip(new, parts))
    return FluxFormAdvection(*new) if changed else advection


def inflate_halo_size(Hx, Hy, Hz, grid, *tendency_terms):
    """Grids/automatic_halo_sizing.jl:68-83"""
    from .grids import Flat
    H = [Hx, Hy, Hz]
    for term in tendency_terms:
        req = (required_halo_size_x(term), required_halo_size_y(term), required_halo_size_z(term))
        H = [0 if t is Flat else max(r, h) for r, h, t in zip(req, H, grid.topology)]
    return tuple(H)
