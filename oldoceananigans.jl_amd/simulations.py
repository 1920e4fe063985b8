"""Simulation-level pieces that touch the hot path: cell_advection_timescale (Advection/cell_advection_timescale.jl:
13-34), cell_diffusion_timescale (TurbulenceClosures/turbulence_closure_diagnostics.jl:7-47) and TimeStepWizard / new_time_step
(Simulations/time_step_wizard.jl:5-115); hasnan / NaNChecker (Diagnostics/nan_checker.jl);
Simulation / run! / Callback / IterationInterval / TimeInterval (Simulations/simulation.jl, run.jl, callback.jl, Utils/schedules.jl) -- host
logic only, every device operation goes through the model's C-ABI calls."""
import ctypes as C
import math
import time as _time
from collections import OrderedDict

import numpy as np

from . import _lib
from .grids import Flat


def cell_advection_timescale(model):
    """min over cells of 1 / (|u|/Δx + |v|/Δy + |w|/Δz); on a distributed model the minimum over ranks"""
    tau = C.c_double()
    _lib.check(_lib.lib().ocn_model_cell_advection_timescale(model.handle, C.byref(tau)))
    return tau.value


def cell_diffusion_timescale(model):
    """min(Δ² / ν, Δ² / max κ) with Δ the smallest spacing (ThreeDimensionalFormulation); Inf without a closure"""
    closure = getattr(model, "closure", None)
    if closure is None:
        return math.inf
    grid = model.grid.local if hasattr(model.grid, "local") else model.grid
    spacings = [grid.Δxᶜᵃᵃ, grid.Δyᵃᶜᵃ, float(np.min(grid.Δzᵃᵃᶜ[grid.Hz:grid.Hz + grid.Nz]))]
    delta = min(d for d, t in zip(spacings, grid.topology) if t is not Flat)
    from .closures import AnisotropicMinimumDissipation
    if isinstance(closure, AnisotropicMinimumDissipation):
        # the eddy coefficients are fields: Δ² / max(νₑ, κₑ) over the model's diffusivity fields (and over the ranks)
        D = model.diffusivity_fields
        arrays = [D[0].parent()] + [k.parent() for k in D[1]]
        biggest = max(float(np.max(a)) for a in arrays)
        if hasattr(model, "ctx") and hasattr(model.ctx, "allreduce_max"):
            biggest = model.ctx.allreduce_max(biggest)
        with np.errstate(divide="ignore"):
            return float(np.float64(delta ** 2) / np.float64(biggest))
    kappas = list(closure.κ.values()) if isinstance(closure.κ, dict) else [closure.κ]
    max_k = max([float(k) for k in kappas] or [0.0])
    with np.errstate(divide="ignore"):
        return float(min(np.float64(delta ** 2) / np.float64(closure.ν), np.float64(delta ** 2) / np.float64(max_k)))


class CFL:
    """CFL(Δt [, timescale = cell_advection_timescale]) (Diagnostics/cfl.jl:3-25): `cfl(model) = Δt / timescale(model)`; Δt a number or a
    TimeStepWizard-like object with a `Δt` (here: a Simulation)"""

    def __init__(self, Δt, timescale=cell_advection_timescale):
        self.Δt, self.timescale = Δt, timescale

    def __call__(self, model):
        dt = self.Δt.Δt if hasattr(self.Δt, "Δt") else self.Δt
        return dt / self.timescale(model)


def AdvectiveCFL(Δt):
    """AdvectiveCFL(Δt) = CFL(Δt, cell_advection_timescale) (cfl.jl:27-50)"""
    return CFL(Δt, cell_advection_timescale)


def DiffusiveCFL(Δt):
    """DiffusiveCFL(Δt) = CFL(Δt, cell_diffusion_timescale) (cfl.jl:52-79)"""
    return CFL(Δt, cell_diffusion_timescale)


class TimeStepWizard:
    """TimeStepWizard(cfl = 0.2, diffusive_cfl = Inf, max_change = 1.1, min_change = 0.5, max_Δt = Inf, min_Δt = 0)"""

    def __init__(self, cfl=0.2, diffusive_cfl=math.inf, max_change=1.1, min_change=0.5, max_Δt=math.inf, min_Δt=0.0):
        if min_change >= 1:
            raise ValueError(f"min_change must be < 1. You provided min_change = {min_change}.")
        if max_change <= 1:
            raise ValueError(f"max_change must be > 1. You provided max_change = {max_change}.")
        self.cfl, self.diffusive_cfl = float(cfl), float(diffusive_cfl)
        self.max_change, self.min_change, self.max_Δt, self.min_Δt = float(max_change), float(min_change), float(max_Δt), float(min_Δt)

    def __call__(self, simulation):
        """(wizard::TimeStepWizard)(simulation) = simulation.Δt = new_time_step(simulation.Δt, wizard, simulation.model)"""
        simulation.Δt = new_time_step(simulation.Δt, self, simulation.model)


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
    if hasattr(obj, "velocities"):
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
    return NaNChecker({"u": model.velocities.u})


# ----------------------------------------------------------------------------------------------------------------------
# schedules, callbacks, Simulation (host logic)
# ----------------------------------------------------------------------------------------------------------------------
class IterationInterval:
    """IterationInterval(interval; offset = 0): actuates when (iteration - offset) % interval == 0 (Utils/schedules.jl:120-123)"""

    def __init__(self, interval, offset=0):
        self.interval, self.offset = int(interval), int(offset)

    def initialize(self, model):
        self(model)
        return True

    def __call__(self, model):
        return (model.clock.iteration - self.offset) % self.interval == 0

    def aligned_time_step(self, clock, Δt):
        return Δt


class TimeInterval:
    """TimeInterval(interval): actuates every `interval` of model time; time steps are shortened to land on it (schedules.jl:30-93)"""

    def __init__(self, interval):
        self.interval, self.first_actuation_time, self.actuations = float(interval), 0.0, 0

    def initialize(self, model):
        self.first_actuation_time, self.actuations = model.clock.time, 0
        return True

    def next_actuation_time(self):
        return self.first_actuation_time + (self.actuations + 1) * self.interval

    def __call__(self, model):
        if model.clock.time >= self.next_actuation_time():
            self.actuations += 1
            return True
        return False

    def aligned_time_step(self, clock, Δt):
        return min(Δt, self.next_actuation_time() - clock.time)


class Callback:
    """Callback(func, schedule = IterationInterval(1); parameters = nothing): func(sim) or func(sim, parameters) (callback.jl)"""

    def __init__(self, func, schedule=None, parameters=None):
        self.func, self.schedule, self.parameters = func, schedule or IterationInterval(1), parameters

    def __call__(self, sim):
        return self.func(sim) if self.parameters is None else self.func(sim, self.parameters)


def stop_iteration_exceeded(sim):
    if sim.model.clock.iteration >= sim.stop_iteration:
        sim.running = False


def stop_time_exceeded(sim):
    if sim.model.clock.time >= sim.stop_time:
        sim.running = False


def wall_time_limit_exceeded(sim):
    if sim.run_wall_time >= sim.wall_time_limit:
        sim.running = False


class Simulation:
    """Simulation(model; Δt, stop_iteration = Inf, stop_time = Inf, wall_time_limit = Inf, align_time_step = true,
    minimum_relative_step = 0) (Simulations/simulation.jl:67-121). `callbacks` is an ordered dict of Callback; the three stop
    criteria and the default NaN checker (every 100 iterations) are installed like the reference does."""

    def __init__(self, model, Δt, stop_iteration=math.inf, stop_time=math.inf, wall_time_limit=math.inf, align_time_step=True,
                 minimum_relative_step=0.0, verbose=False):
        self.model, self.Δt = model, float(Δt)
        self.stop_iteration, self.stop_time, self.wall_time_limit = float(stop_iteration), float(stop_time), float(wall_time_limit)
        self.align_time_step, self.minimum_relative_step, self.verbose = bool(align_time_step), float(minimum_relative_step), verbose
        self.run_wall_time, self.initialized, self.running = 0.0, False, False
        self.callbacks = OrderedDict()
        self.callbacks["stop_time_exceeded"] = Callback(stop_time_exceeded)
        self.callbacks["stop_iteration_exceeded"] = Callback(stop_iteration_exceeded)
        self.callbacks["wall_time_limit_exceeded"] = Callback(wall_time_limit_exceeded)
        self.callbacks["nan_checker"] = Callback(default_nan_checker(model), IterationInterval(100))

    # -- one time-step ---------------------------------------------------------------------------------------------
    def aligned_time_step(self, Δt):
        """aligned_time_step(sim, Δt) (run.jl:43-59): callbacks on a TimeInterval first, then the stop time (which wins)"""
        clock = self.model.clock
        aligned = Δt
        for cb in self.callbacks.values():
            aligned = cb.schedule.aligned_time_step(clock, aligned)
        aligned = min(aligned, self.stop_time - clock.time)
        return Δt if aligned <= 0 else aligned

    def initialize(self):
        """initialize!(sim) (run.jl:196-251): update_state!, schedules, callbacks at iteration 0"""
        from .models import update_state
        if hasattr(self.model, "handle"):
            update_state(self.model, True)
        for cb in self.callbacks.values():
            cb.schedule.initialize(self.model)
        if self.model.clock.iteration == 0:
            for cb in self.callbacks.values():
                cb(self)
        self.initialized = True

    def time_step(self, Δt=None):
        """time_step!(sim) / time_step!(sim, Δt) (run.jl:113-174)"""
        from .models import time_step as model_time_step
        start = _time.perf_counter()
        if Δt is not None:
            self.Δt, self.align_time_step = float(Δt), False
        Δt = self.aligned_time_step(self.Δt) if self.align_time_step else self.Δt
        if not self.initialized:
            self.initialize()
        if Δt < self.minimum_relative_step * self.Δt:
            raise NotImplementedError("skipping a tiny aligned time step needs a writable clock (minimum_relative_step > 0)")
        model_time_step(self.model, Δt)
        for cb in self.callbacks.values():
            if cb.schedule(self.model):
                cb(self)
        self.run_wall_time += _time.perf_counter() - start

    def reset(self):
        """reset!(sim) (simulation.jl:203-213)"""
        _lib.check(_lib.lib().ocn_model_reset(self.model.handle))
        self.stop_iteration = self.stop_time = self.wall_time_limit = math.inf
        self.run_wall_time, self.initialized, self.running = 0.0, False, True


def run(sim):
    """run!(simulation) (run.jl:97-111): step until a stop criterion clears `running`"""
    sim.initialized, sim.running, sim.run_wall_time = False, True, 0.0
    while sim.running:
        sim.time_step()


def reset(sim):
    sim.reset()
