"""Buoyancy formulations on the accelerated path, gravity along -z (reference: src/BuoyancyFormulations/)."""


class BuoyancyTracer:
    """BuoyancyTracer(): the tracer `b` is the buoyancy (buoyancy_tracer.jl:1-12)"""
    required_tracers = ("b",)

    def __repr__(self):
        return "BuoyancyTracer()"


class LinearEquationOfState:
    """LinearEquationOfState(thermal_expansion = 1.67e-4, haline_contraction = 7.80e-4) (linear_equation_of_state.jl:39-41)"""

    def __init__(self, thermal_expansion=1.67e-4, haline_contraction=7.80e-4):
        self.thermal_expansion, self.haline_contraction = float(thermal_expansion), float(haline_contraction)


class SeawaterBuoyancy:
    """SeawaterBuoyancy(equation_of_state = LinearEquationOfState(), gravitational_acceleration = g_Earth): b = g (α T - β S)
    (seawater_buoyancy.jl, linear_equation_of_state.jl:71-73). Nonlinear equations of state (TEOS-10) are third-party code absent
    from the reference tree and stay out of scope."""
    required_tracers = ("T", "S")

    def __init__(self, equation_of_state=None, gravitational_acceleration=9.80665):
        self.equation_of_state = equation_of_state if equation_of_state is not None else LinearEquationOfState()
        if not isinstance(self.equation_of_state, LinearEquationOfState):
            raise NotImplementedError("only LinearEquationOfState is on the accelerated path")
        self.gravitational_acceleration = float(gravitational_acceleration)

    def __repr__(self):
        e = self.equation_of_state
        return (f"SeawaterBuoyancy(g={self.gravitational_acceleration}, LinearEquationOfState(α={e.thermal_expansion}, "
                f"β={e.haline_contraction}))")


def sind(x):
    """sin of an angle in degrees as Julia's `sind` returns it: exact at multiples of 30 and 45 degrees, correctly rounded elsewhere
    (math.sin(math.radians(45)) is one ulp below sind(45) = 0.7071067811865476)"""
    from decimal import Decimal, getcontext
    from fractions import Fraction
    getcontext().prec = 60
    r = Fraction(float(x)) % 360
    sign = 1
    if r >= 180:
        r, sign = r - 180, -1
    if r > 90:
        r = 180 - r
    if r == 0:
        return 0.0 * sign
    if r == 90:
        return 1.0 * sign
    if r == 30:
        return 0.5 * sign
    pi = Decimal("3.14159265358979323846264338327950288419716939937510582097494459")
    t = Decimal(r.numerator) / Decimal(r.denominator) * pi / 180
    term, total, n = t, t, 1
    while abs(term) > Decimal(10) ** -55:
        term = -term * t * t / ((2 * n) * (2 * n + 1))
        total += term
        n += 1
    return float(total) * sign


class FPlane:
    """FPlane(f = ...) | FPlane(rotation_rate = Ω, latitude = φ): f = 2 Ω sind(φ) (Coriolis/f_plane.jl:13-44)"""

    def __init__(self, f=None, rotation_rate=7.292115e-5, latitude=None):
        import math
        if (f is None) == (latitude is None):
            raise ValueError("Either both keywords rotation_rate and latitude must be specified, *or* only f must be specified.")
        self.f = float(f) if f is not None else 2 * rotation_rate * sind(latitude)      # f_plane.jl:38: 2rotation_rate * sind(latitude)

    def __repr__(self):
        return f"FPlane(f={self.f})"
