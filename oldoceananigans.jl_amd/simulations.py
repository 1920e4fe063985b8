"""Simulation-level pieces that touch the hot path: cell_advection_timescale (Advection/cell_advection_timescale.jl:
13-34), cell_diffusion_timescale (TurbulenceClosures/turbulence_closure_diagnostics.jl:7-47) and TimeStepWizard / new_time_step
(Simulations/time_step_wizard.jl:5-115); hasnan / NaNChecker (Diagnostics/nan_checker.jl)."""
import ctypes as C
import math

import numpy as np

from . import _lib
from .grids import Flat


def cell_advection_timescale(model):
    """min over cells of 1 / (|u|/Δx + |v|/Δy + |w|/Δz); on a distributed model the minimum over ranks"""
    tau = C.c_double()
    if hasattr(model, "backend"):                     # DistributedNonhydrostaticModel
        b = model.backend
        _lib.check(_lib.lib().ocn_cell_advection_timescale(model.grid.local.handle, b.U[0].data, b.U[1].data, b.U[2].data, C.byref(tau)))
        return -model.ctx.allreduce_max(-tau.value)
    _lib.check(_lib.lib().ocn_model_cell_advection_timescale(model.handle, C.byref(tau)))
    return tau.value


def cell_diffusion_timescale(model):
    """min(Δ² / ν, Δ² / max κ) with Δ the smallest spacing (ThreeDimensionalFormulation); Inf without a closure"""
    closure = getattr(model, "closure", None) or getattr(getattr(model, "backend", None), "closure", None)
    if closure is None:
        return math.inf
    grid = model.grid.local if hasattr(model.grid, "local") else model.grid
    spacings = [grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, float(np.min(grid.Δzᵃᵃᶜ[grid.Hz:grid.Hz + grid.Nz]))]
    delta = min(d for d, t in zip(spacings, grid.topology) if t is not Flat)
    kappas = list(closure.κ.values()) if isinstance(closure.κ, dict) else [closure.κ]
    max_k = max([float(k) for k in kappas] or [0.0])
    with np.errstate(divide="ignore"):
        return float(min(np.float64(delta ** 2) / np.float64(closure.ν), np.float64(delta ** 2) / np.float64(max_k)))


class TimeStepWizard:
    """TimeStepWizard(cfl = 0.2, diffusive_cfl = Inf, max_change = 1.1, min_change = 0.5, max_Δt = Inf, min_Δt = 0)"""

    def __init__(self, cfl=0.2, diffusive_cfl=math.inf, max_change=1.1, min_change=0.5, max_Δt=math.inf, min_Δt=0.0):
        if min_change >= 1:
            raise ValueError(f"min_change must be < 1. You provided min_change = {min_change}.")
        if max_change <= 1:
            raise ValueError(f"max_change must be > 1. You provided max_change = {max_change}.")
        self.cfl, self.diffusive_cfl = float(cfl), float(diffusive_cfl)
        self.max_change, self.min_change, self.max_Δt, self.min_Δt = float(max_change), float(min_change), float(max_Δt), float(min_Δt)


def new_time_step(old_Δt, wizard, model):
    """new_time_step(old_Δt, wizard, model) (time_step_wizard.jl:101-115)"""
    advective = wizard.cfl * cell_advection_timescale(model)
    diffusive = wizard.diffusive_cfl * cell_diffusion_timescale(model) if math.isfinite(wizard.diffusive_cfl) else math.inf
    new = min(advective, diffusive)
    new = min(wizard.max_change * old_Δt, new)
    new = max(wizard.min_change * old_Δt, new)
    return min(max(new, wizard.min_Δt), wizard.max_Δt)


def hasnan(obj):
    """hasnan(field) = any(isnan, parent(field)); hasnan(model) checks the first prognostic field (nan_checker.jl:32-33)"""
    field = obj
    if hasattr(obj, "backend"):
        field = obj.backend.U[0]
    elif hasattr(obj, "velocities"):
        field = obj.velocities.u
    r = C.c_int()
    _lib.check(_lib.lib().ocn_hasnan(field.data, field.nbytes // 8, C.byref(r)))
    found = bool(r.value)
    if hasattr(obj, "ctx") and obj.ctx.world > 1:
        found = obj.ctx.allreduce_max(float(found)) > 0
    return found


class NaNChecker:
    """NaNChecker(; fields, erroring = false): called with a simulation (any object with `.running` and `.model.clock`), stops
    it -- or raises when erroring -- if one of the fields holds a NaN (nan_checker.jl:35-53)"""

    def __init__(self, fields, erroring=False):
        self.fields, self.erroring = dict(fields), bool(erroring)

    def __call__(self, simulation):
        for name, field in self.fields.items():
            if hasnan(field):
                simulation.running = False
                clock = simulation.model.clock
                msg = f"time = {clock.time}, iteration = {clock.iteration}: NaN found in field {name}."
                if self.erroring:
                    raise RuntimeError(msg + " Aborting simulation.")
                print("[ Info: " + msg + " Stopping simulation.")


def default_nan_checker(model):
    """NaNChecker on the first prognostic field, u (Models/Models.jl:173-184)"""
    return NaNChecker({"u": model.velocities.u if hasattr(model, "velocities") else model.backend.U[0]})
