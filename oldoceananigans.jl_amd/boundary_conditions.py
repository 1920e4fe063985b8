"""Boundary conditions with constant or array values (reference: src/BoundaryConditions/boundary_condition.jl,
boundary_condition_classifications.jl, field_boundary_conditions.jl). Function- and field-valued conditions are outside the
accelerated path."""
import ctypes as C

import numpy as np

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
        self.array = None
        if isinstance(condition, np.ndarray) or (isinstance(condition, (list, tuple)) and np.ndim(condition) == 2):
            # getbc(condition::AbstractArray, i, j, grid, args...) = condition[i, j] (boundary_condition.jl:164): a 2-D array over the
            # interior extents of the two tangential directions, uploaded once and handed to the library as a device pointer
            self.array = np.asfortranarray(condition, dtype=np.float64)
            if self.array.ndim != 2:
                raise ValueError("an array-valued boundary condition is a 2-D array over the two tangential directions")
            self._device = None
            condition = 0.0
        if callable(condition) or not isinstance(condition, (int, float)):
            raise NotImplementedError("only constant (Number) or array boundary conditions and LinearFieldFlux are on the accelerated path")
        self.classification, self.condition = classification, float(condition)

    def device_array(self, expected_shape=None):
        """device copy of an array-valued condition (None for a number), created on first use and freed with the condition"""
        if self.array is None:
            return None
        if expected_shape is not None and tuple(self.array.shape) != tuple(expected_shape):
            raise ValueError(f"boundary condition array has shape {self.array.shape}, the boundary has {tuple(expected_shape)} points")
        if self._device is None:
            p = C.c_void_p()
            _lib.check(_lib.lib().ocn_malloc(C.byref(p), self.array.nbytes))
            _lib.check(_lib.lib().ocn_memcpy_h2d(p, self.array.ctypes.data_as(C.c_void_p), self.array.nbytes))
            self._device = p
        return self._device

    def __del__(self):
        try:
            if getattr(self, "_device", None) is not None:
                _lib.lib().ocn_free(self._device)
        except Exception:
            pass

    def __repr__(self):
        if self.linear:
            return f"FluxBoundaryCondition: {self.linear[0]} + {self.linear[1]} * {self.linear[2]}"
        if self.array is not None:
            return f"{self.classification}BoundaryCondition: {self.array.shape[0]}×{self.array.shape[1]} Array{{Float64, 2}}"
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

    def c_array(self, grid=None):
        arr = (_lib.BC * 6)()
        for s, bc in self.sides.items():
            q = SIDES.index(s)
            arr[q].kind = KINDS[bc.classification]
            arr[q].value = bc.condition
            if bc.array is not None:
                arr[q].array = bc.device_array(_tangential_shape(grid, q) if grid is not None else None)
        arr._keep = self                    # the device arrays live as long as the conditions
        return arr


def _tangential_shape(grid, side):
    """interior extents of the two tangential directions of side 0..5 (west, east, south, north, bottom, top), x before y before z"""
    N = grid.size
    d = side // 2
    return (N[1], N[2]) if d == 0 else ((N[0], N[2]) if d == 1 else (N[0], N[1]))


def bc_table(fields_bcs, grid=None):
    """list of FieldBoundaryConditions | None, one per field -> ocn_bc_t[n][6]"""
    table = ((_lib.BC * 6) * len(fields_bcs))()
    table._keep = list(fields_bcs)
    for f, fb in enumerate(fields_bcs):
        if fb is not None:
            row = fb.c_array(grid)
            for s in range(6):
                table[f][s].kind, table[f][s].value, table[f][s].array = row[s].kind, row[s].value, row[s].array
    return table


def compute_flux_bcs(G, bcs):
    """compute_x_bcs! / compute_y_bcs! / compute_z_bcs! (compute_flux_bcs.jl:12-163) on the tendency field G"""
    loc = (C.c_int * 3)(*[1 if l.__name__ == "Face" else 0 for l in G.loc])
    _lib.check(_lib.lib().ocn_compute_flux_bcs(G.grid.handle, G.data, loc, bcs.c_array(G.grid)))
