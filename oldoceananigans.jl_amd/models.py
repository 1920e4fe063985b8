"""NonhydrostaticModel + RK3 time stepping (reference: src/Models/NonhydrostaticModels/nonhydrostatic_model.jl:115-244,
src/TimeSteppers/runge_kutta_3.jl:93-170). The whole time-step runs inside libocn_mi355x.so (`ocn_model_time_step`);
this module is the host-side mirror of the reference's API."""
import ctypes as C
import weakref
from collections import namedtuple

from . import _lib
from .advection import WENO
from .fields import Field
from .grids import Center, Face


class Clock:
    """TimeSteppers/clock.jl:39-45 (read-only view of the library's clock). Holds a WEAK reference to its model: no reference
    cycle, so a model is destroyed (and its FFT plans released) as soon as the last user reference goes away."""

    def __init__(self, model):
        self._model = weakref.ref(model)

    def _get(self):
        model = self._model()
        if model is None or model.handle is None:
            raise _lib.OcnError("the model of this clock has been destroyed")
        t, it, st, ldt, lsdt = C.c_double(), C.c_int64(), C.c_int(), C.c_double(), C.c_double()
        _lib.check(_lib.lib().ocn_model_clock(model.handle, C.byref(t), C.byref(it), C.byref(st), C.byref(ldt),
                                              C.byref(lsdt)))
        return t.value, it.value, st.value, ldt.value, lsdt.value

    time = property(lambda self: self._get()[0])
    iteration = property(lambda self: self._get()[1])
    stage = property(lambda self: self._get()[2])
    last_Δt = property(lambda self: self._get()[3])
    last_stage_Δt = property(lambda self: self._get()[4])


class NonhydrostaticModel:
    """NonhydrostaticModel(; grid, advection=WENO(), tracers=(:T, :S), timestepper=:RungeKutta3) with
    coriolis / buoyancy / closure / forcing = nothing -- the configuration BASELINE.json benchmarks."""

    def __init__(self, grid, advection=None, tracers=("T", "S"), timestepper="RungeKutta3", buoyancy=None, coriolis=None,
                 closure=None, forcing=None, boundary_conditions=None):
        if advection is None:
            advection = WENO()
        if not (isinstance(advection, WENO) and advection.order == 5 and advection.bounds is None):
            raise NotImplementedError("only advection = WENO(order=5) without bounds is on the accelerated hot path")
        timestepper = str(timestepper).lstrip(":")
        if timestepper not in ("RungeKutta3", "QuasiAdamsBashforth2"):
            raise NotImplementedError("timestepper must be :RungeKutta3 (hot path) or :QuasiAdamsBashforth2 (SURVEY.md 8f.1)")
        self.timestepper, self.χ = timestepper, 0.1          # QuasiAdamsBashforth2TimeStepper(χ = 0.1)
        if forcing is not None:
            raise NotImplementedError("forcing != nothing is outside the accelerated hot path (SURVEY.md 8f)")
        from .buoyancy import BuoyancyTracer, FPlane, SeawaterBuoyancy
        if coriolis is not None and not isinstance(coriolis, FPlane):
            raise NotImplementedError("coriolis must be nothing or FPlane(f)")
        self.coriolis = coriolis
        if buoyancy is not None and not isinstance(buoyancy, (BuoyancyTracer, SeawaterBuoyancy)):
            raise NotImplementedError("buoyancy must be nothing, BuoyancyTracer() or SeawaterBuoyancy(LinearEquationOfState)")
        self.buoyancy = buoyancy
        from .closures import AnisotropicMinimumDissipation, ScalarDiffusivity
        if closure is not None and not isinstance(closure, (ScalarDiffusivity, AnisotropicMinimumDissipation)):
            raise NotImplementedError("only closure = nothing | ScalarDiffusivity(ν, κ) | AnisotropicMinimumDissipation(C, Cν, Cκ) is on "
                                      "the accelerated path (SURVEY.md 8f)")
        self.closure = closure
        # "Adjust advection scheme to be valid on a particular grid size" and "Adjust halos when the advection scheme or turbulence
        # closure requires it" (nonhydrostatic_model.jl:176-184). The library derives the same per-direction schemes from the grid
        # size (ocn_grid_create), so only the descriptor and the halo are settled here.
        from .advection import adapt_advection_order, inflate_halo_size
        from .grids import with_halo
        advection = adapt_advection_order(advection, grid)
        required = inflate_halo_size(*grid.halo_size, grid, advection, closure)
        if any(u < r for u, r in zip(grid.halo_size, required)):
            import warnings
            warnings.warn(f"Inflating model grid halo size to {required} and recreating grid. The model grid will be different from the "
                          f"input grid. To avoid this warning, pass halo={required} when constructing the grid.")
            grid = with_halo(required, grid)
        self.grid, self.advection = grid, advection
        self.tracer_names = tuple(str(t) for t in (tracers if isinstance(tracers, (tuple, list)) else (tracers,)))
        self.handle = self._create_handle(grid, len(self.tracer_names))
        self.clock = Clock(self)
        V = namedtuple("Velocities", "u v w")
        self.velocities = V(self._field("u"), self._field("v"), self._field("w"))
        T = namedtuple("Tracers", self.tracer_names) if self.tracer_names else tuple
        self.tracers = T(*[self._field("c%d" % n) for n in range(len(self.tracer_names))])
        if buoyancy is not None:
            # validate_buoyancy / tracernames (nonhydrostatic_model.jl:166-170): the formulation's tracers must exist
            missing = [t for t in buoyancy.required_tracers if t not in self.tracer_names]
            if missing:
                raise ValueError(f"{buoyancy!r} requires the tracers {missing}")
            idx = [self.tracer_names.index(t) for t in buoyancy.required_tracers]
            if isinstance(buoyancy, BuoyancyTracer):
                _lib.check(_lib.lib().ocn_model_set_buoyancy(self.handle, 1, idx[0], 0, 0.0, 0.0, 0.0))
            else:
                e = buoyancy.equation_of_state
                _lib.check(_lib.lib().ocn_model_set_buoyancy(self.handle, 2, idx[0], idx[1], buoyancy.gravitational_acceleration,
                                                             e.thermal_expansion, e.haline_contraction))
            P = namedtuple("Pressures", "pNHS pHY")          # pHY′: the hydrostatic pressure anomaly (nonhydrostatic_model.jl:144-158)
            self.pressures = P(self._field("p"), self._field("pHY"))
        else:
            P = namedtuple("Pressures", "pNHS")
            self.pressures = P(self._field("p"))
        if coriolis is not None:
            _lib.check(_lib.lib().ocn_model_set_coriolis(self.handle, 1, coriolis.f))
        self.diffusivity_fields = None
        if isinstance(closure, AnisotropicMinimumDissipation):
            self._kappa, kp = closure.Ckappa_array(self.tracer_names)
            _lib.check(_lib.lib().ocn_model_set_amd(self.handle, closure.Cν, kp))
            # build_diffusivity_fields (anisotropic_minimum_dissipation.jl:339-352). Python normalises identifiers (NFKC): the
            # attribute written `.νₑ` in source is looked up as "νe", so the tuple's field names are the normalised spellings
            import unicodedata
            D = namedtuple("DiffusivityFields", [unicodedata.normalize("NFKC", n) for n in ("νₑ", "κₑ")])
            K = namedtuple("EddyDiffusivities", self.tracer_names) if self.tracer_names else tuple
            self.diffusivity_fields = D(self._field("nu_e"), K(*[self._field("kappa_e%d" % n) for n in range(len(self.tracer_names))]))
        elif closure is not None:
            self._kappa, kp = closure.kappa_array(self.tracer_names)
            _lib.check(_lib.lib().ocn_model_set_closure(self.handle, closure.ν, kp))
        # boundary_conditions = (u = FieldBoundaryConditions(top = FluxBoundaryCondition(Q)), ...) (nonhydrostatic_model.jl:
        # 163-190): constant Flux / Value / Gradient / Open conditions on Bounded sides
        self.boundary_conditions = dict(boundary_conditions or {})
        from .boundary_conditions import KINDS, SIDES
        import unicodedata
        targets = []                                     # (library field name, FieldBoundaryConditions)
        for name, fbcs in self.boundary_conditions.items():
            key = unicodedata.normalize("NFKC", name)
            if key in ("νe", "nu_e", "κe", "kappa_e"):
                # boundary_conditions = (νₑ = FieldBoundaryConditions(...), κₑ = (b = FieldBoundaryConditions(...),)): the diffusivity
                # fields of an LES closure (anisotropic_minimum_dissipation.jl:339-352)
                if self.diffusivity_fields is None:
                    raise ValueError(f"boundary conditions given for {name}, but the closure has no diffusivity fields")
                if key in ("νe", "nu_e"):
                    targets.append(("nu_e", fbcs))
                else:
                    for tracer, tb in dict(fbcs).items():
                        if tracer not in self.tracer_names:
                            raise ValueError(f"κₑ boundary conditions given for {tracer}, which is not a tracer of the model")
                        targets.append(("kappa_e%d" % self.tracer_names.index(tracer), tb))
                continue
            if name not in ("u", "v", "w") + self.tracer_names:
                raise ValueError(f"boundary conditions given for {name}, which is not a velocity or tracer of the model")
            targets.append((self._cname(name), fbcs))
        for cname, fbcs in targets:
            for side, bc in fbcs.sides.items():
                if bc.linear is not None:
                    a, b, dep = bc.linear
                    _lib.check(_lib.lib().ocn_model_set_linear_flux_bc(self.handle, cname.encode(), SIDES.index(side), a, b,
                                                                       self._cname(dep).encode()))
                    continue
                if bc.array is not None:
                    from .boundary_conditions import _tangential_shape
                    dev = bc.device_array(_tangential_shape(self.grid, SIDES.index(side)))      # borrowed by the library: `bc` is kept
                    _lib.check(_lib.lib().ocn_model_set_boundary_condition_array(self.handle, cname.encode(), SIDES.index(side),
                                                                                 KINDS[bc.classification], dev))
                    continue
                _lib.check(_lib.lib().ocn_model_set_boundary_condition(self.handle, cname.encode(), SIDES.index(side),
                                                                       KINDS[bc.classification], bc.condition))

    def _create_handle(self, grid, ntracers):
        h = C.c_void_p()
        _lib.check(_lib.lib().ocn_model_create(C.byref(h), grid.handle, ntracers))
        return h

    @property
    def architecture(self):
        return self.grid.architecture

    def _field(self, cname):
        p, loc = C.c_void_p(), (C.c_int * 3)()
        _lib.check(_lib.lib().ocn_model_field(self.handle, cname.encode(), C.byref(p), loc))
        return Field(tuple(Face if l else Center for l in loc), self.grid, data=p)   # a VIEW: valid while the model lives

    def tendency(self, name, previous=False):
        """timestepper.Gⁿ[name] / G⁻[name]; pointers are stable at time-step boundaries."""
        return self._field(("M" if previous else "G") + self._cname(name))

    def _cname(self, name):
        if name in ("u", "v", "w"):
            return name
        if name in self.tracer_names:
            return "c%d" % self.tracer_names.index(name)
        raise ValueError(f"name {name} not found in model.velocities or model.tracers.")

    def fields(self):
        d = dict(zip("uvw", self.velocities))
        d.update(dict(zip(self.tracer_names, self.tracers)))
        return d

    def set_option(self, key, value):
        _lib.check(_lib.lib().ocn_model_set_option(self.handle, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int()
        _lib.check(_lib.lib().ocn_model_get_option(self.handle, key.encode(), C.byref(v)))
        return v.value

    def profile_read(self):
        """(total ms, count) of the event-timed tendency evaluations since the last read"""
        ms, n = C.c_double(), C.c_int()
        _lib.check(_lib.lib().ocn_model_profile_read(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        """release the model's device memory and FFT plans now (also called by the destructor)"""
        if getattr(self, "handle", None) is not None:
            _lib.lib().ocn_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_model(model, enforce_incompressibility=True, **kwargs):
    """set!(model; enforce_incompressibility=true, kwargs...) (set_nonhydrostatic_model.jl:33-60)"""
    flds = model.fields()
    for name, value in kwargs.items():
        if name not in flds:
            raise ValueError(f"name {name} not found in model.velocities or model.tracers.")
        flds[name].set(value)
    _lib.check(_lib.lib().ocn_model_set_finalize(model.handle, int(enforce_incompressibility)))


def update_state(model, compute_tendencies=True):
    """update_state!(model; compute_tendencies) (update_nonhydrostatic_model_state.jl:20-56)"""
    _lib.check(_lib.lib().ocn_model_update_state(model.handle, int(compute_tendencies)))


def time_step(model, Δt, euler=False):
    """time_step!(model, Δt) (runge_kutta_3.jl:93-170; quasi_adams_bashforth_2.jl:74-123 when the model's timestepper is
    :QuasiAdamsBashforth2 -- `euler` as in the reference)"""
    if getattr(model, "timestepper", "RungeKutta3") == "QuasiAdamsBashforth2":
        _lib.check(_lib.lib().ocn_model_time_step_ab2(model.handle, float(Δt), float(model.χ), int(euler)))
    else:
        _lib.check(_lib.lib().ocn_model_time_step(model.handle, float(Δt)))


def max_abs_divergence(model):
    if hasattr(model, "ctx") and hasattr(model, "max_abs_divergence"):       # partitioned model: global maximum
        return model.max_abs_divergence()
    v = C.c_double()
    _lib.check(_lib.lib().ocn_model_max_abs_divergence(model.handle, C.byref(v)))
    return v.value
