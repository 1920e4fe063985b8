"""Boundary conditions with constant values (reference: src/BoundaryConditions/boundary_condition.jl,
boundary_condition_classifications.jl, field_boundary_conditions.jl). Function- and field-valued conditions are outside the
accelerated path."""
import ctypes as C

from . import _lib

SIDES = ("west", "east", "south", "north", "bottom", "top")
KINDS = {"Default": 0, "Flux": 1, "Value": 2, "Gradient": 3, "Open": 4}


class LinearFieldFlux:
    """the boundary function (ξ, η, t, φ, p) -> a + b φ of one field dependency φ: the family of field-dependent Flux conditions that
    crosses the C ABI (ocn_model_set_linear_flux_bc). Jˢ(x, y, t, S, rate) = -rate S of examples/ocean_wind_mixing_and_convection.jl:
    125 is LinearFieldFlux(b=-rate)."""

    def __init__(self, a=0.0, b=0.0):
        self.a, self.b = float(a), float(b)


class BoundaryCondition:
    """BoundaryCondition(classification, condition::Number); Flux conditions also take a LinearFieldFlux with field_dependencies"""

    def __init__(self, classification, condition=0.0, field_dependencies=None):
        if classification not in KINDS:
            raise ValueError(f"unknown boundary condition classification {classification}")
        self.linear = None
        if isinstance(condition, LinearFieldFlux):
            deps = (field_dependencies,) if isinstance(field_dependencies, str) else tuple(field_dependencies or ())
            if classification != "Flux" or len(deps) != 1:
                raise NotImplementedError("a LinearFieldFlux is a Flux condition with exactly one field dependency")
            self.linear = (condition.a, condition.b, str(deps[0]).lstrip(":"))
            condition = 0.0
        if callable(condition) or not isinstance(condition, (int, float)):
            raise NotImplementedError("only constant (Number) boundary conditions and LinearFieldFlux are on the accelerated path")
        self.classification, self.condition = classification, float(condition)

    def __repr__(self):
        if self.linear:
            return f"FluxBoundaryCondition: {self.linear[0]} + {self.linear[1]} * {self.linear[2]}"
        return f"{self.classification}BoundaryCondition: {self.condition}"


def FluxBoundaryCondition(value, field_dependencies=None, parameters=None):
    """FluxBoundaryCondition(Number) | FluxBoundaryCondition(LinearFieldFlux(a, b), field_dependencies = :φ)"""
    return BoundaryCondition("Flux", value, field_dependencies)


def ValueBoundaryCondition(value):
    return BoundaryCondition("Value", value)


def GradientBoundaryCondition(value):
    return BoundaryCondition("Gradient", value)


def OpenBoundaryCondition(value):
    return BoundaryCondition("Open", value)


class FieldBoundaryConditions:
    """FieldBoundaryConditions(; west, east, south, north, bottom, top): unspecified sides keep the defaults of
    field_boundary_conditions.jl:15-25 (Periodic -> periodic, Bounded + Center -> no flux, Bounded + Face -> impenetrable)"""

    def __init__(self, **sides):
        for s, bc in sides.items():
            if s not in SIDES:
                raise ValueError(f"unknown side {s}; expected one of {SIDES}")
            if bc is not None and not isinstance(bc, BoundaryCondition):
                raise TypeError(f"{s} must be a BoundaryCondition")
        self.sides = {s: bc for s, bc in sides.items() if bc is not None}

    def c_array(self):
        arr = (_lib.BC * 6)()
        for s, bc in self.sides.items():
            arr[SIDES.index(s)].kind = KINDS[bc.classification]
            arr[SIDES.index(s)].value = bc.condition
        return arr


def bc_table(fields_bcs):
    """list of FieldBoundaryConditions | None, one per field -> ocn_bc_t[n][6]"""
    table = ((_lib.BC * 6) * len(fields_bcs))()
    for f, fb in enumerate(fields_bcs):
        if fb is not None:
            row = fb.c_array()
            for s in range(6):
                table[f][s].kind, table[f][s].value = row[s].kind, row[s].value
    return table


def compute_flux_bcs(G, bcs):
    """compute_x_bcs! / compute_y_bcs! / compute_z_bcs! (compute_flux_bcs.jl:12-163) on the tendency field G"""
    loc = (C.c_int * 3)(*[1 if l.__name__ == "Face" else 0 for l in G.loc])
    _lib.check(_lib.lib().ocn_compute_flux_bcs(G.grid.handle, G.data, loc, bcs.c_array()))
