"""Raw kernel entry points on borrowed Fields: what `launch!(arch, grid, workspec, kernel!, args...)` runs for each
hot-path kernel of the reference."""
import ctypes as C

from . import _lib
from .fields import _loc_array, _ptr_array


def _range(r):
    return None if r is None else (C.c_int * 6)(*[int(x) for x in r])


def compute_Gu(grid, u, v, w, Gu, kernel_parameters=None):
    _lib.check(_lib.lib().ocn_compute_Gu(grid.handle, u.data, v.data, w.data, Gu.data, _range(kernel_parameters)))


def compute_Gv(grid, u, v, w, Gv, kernel_parameters=None):
    _lib.check(_lib.lib().ocn_compute_Gv(grid.handle, u.data, v.data, w.data, Gv.data, _range(kernel_parameters)))


def compute_Gw(grid, u, v, w, Gw, kernel_parameters=None):
    _lib.check(_lib.lib().ocn_compute_Gw(grid.handle, u.data, v.data, w.data, Gw.data, _range(kernel_parameters)))


def compute_Gc(grid, u, v, w, c, Gc, kernel_parameters=None):
    _lib.check(_lib.lib().ocn_compute_Gc(grid.handle, u.data, v.data, w.data, c.data, Gc.data, _range(kernel_parameters)))


def compute_tendencies(grid, u, v, w, tracers, Gu, Gv, Gw, Gc, kernel_parameters=None):
    """compute_interior_tendency_contributions! as one fused flux-sharing pass"""
    tr = _ptr_array(tracers) if tracers else None
    gc = _ptr_array(Gc) if Gc else None
    _lib.check(_lib.lib().ocn_compute_tendencies(grid.handle, u.data, v.data, w.data, tr, len(tracers), Gu.data, Gv.data,
                                                 Gw.data, gc, _range(kernel_parameters)))


def compute_tendencies_and_substep(grid, fields, Gn, next_fields, Gm, Δt, γ, ζ, kernel_parameters=None):
    """tendencies of all prognostic fields (u, v, w, tracers...) + the rk3_substep! of the next stage into `next_fields`"""
    _lib.check(_lib.lib().ocn_compute_tendencies_and_substep(
        grid.handle, _ptr_array(fields), len(fields) - 3, _ptr_array(Gn), _range(kernel_parameters), _ptr_array(next_fields),
        _ptr_array(Gm), float(Δt), float(γ), 0.0 if ζ is None else float(ζ), 0 if ζ is None else 1))


def compute_closure_tendencies(grid, fields, Gn, closure, tracer_names, kernel_parameters=None):
    """adds the ScalarDiffusivity terms (-∂ⱼτᵢⱼ, -∇·q) to tendencies that hold the advective part; fields = u, v, w, tracers..."""
    karr, kp = closure.kappa_array(tracer_names)
    tr, gc = fields[3:], Gn[3:]
    _lib.check(_lib.lib().ocn_compute_closure_tendencies(
        grid.handle, fields[0].data, fields[1].data, fields[2].data, _ptr_array(tr) if tr else None, len(tr), closure.ν, kp,
        Gn[0].data, Gn[1].data, Gn[2].data, _ptr_array(gc) if gc else None, _range(kernel_parameters)))


def compute_amd_diffusivities(grid, closure, tracer_names, fields, νₑ, κₑ, kernel_parameters=None):
    """compute_diffusivities!(…, closure::AnisotropicMinimumDissipation, …) over the interior; fields = u, v, w, tracers... with filled
    halos; fill the halos of νₑ, κₑ afterwards (fill_halo_regions)"""
    karr, kp = closure.Ckappa_array(tracer_names)
    tr = fields[3:]
    _lib.check(_lib.lib().ocn_compute_amd_diffusivities(
        grid.handle, closure.Cν, kp, fields[0].data, fields[1].data, fields[2].data, _ptr_array(tr) if tr else None, len(tr),
        νₑ.data, _ptr_array(κₑ) if κₑ else None, _range(kernel_parameters)))


def compute_closure_tendencies_field(grid, fields, Gn, νₑ, κₑ, kernel_parameters=None):
    """adds -∂ⱼτᵢⱼ, -∇·q with the coefficients read from the ccc arrays νₑ, κₑ[tracer] (halos filled)"""
    tr, gc = fields[3:], Gn[3:]
    _lib.check(_lib.lib().ocn_compute_closure_tendencies_field(
        grid.handle, fields[0].data, fields[1].data, fields[2].data, _ptr_array(tr) if tr else None, len(tr), νₑ.data,
        _ptr_array(κₑ) if κₑ else None, Gn[0].data, Gn[1].data, Gn[2].data, _ptr_array(gc) if gc else None, _range(kernel_parameters)))


def update_hydrostatic_pressure(grid, buoyancy, tracers_by_name, pHY):
    """update_hydrostatic_pressure! (update_hydrostatic_pressure.jl:12-49) for BuoyancyTracer | linear SeawaterBuoyancy"""
    from .buoyancy import BuoyancyTracer
    if isinstance(buoyancy, BuoyancyTracer):
        _lib.check(_lib.lib().ocn_update_hydrostatic_pressure(grid.handle, 1, tracers_by_name["b"].data, None, 0.0, 0.0, 0.0, pHY.data))
    else:
        e = buoyancy.equation_of_state
        _lib.check(_lib.lib().ocn_update_hydrostatic_pressure(grid.handle, 2, tracers_by_name["T"].data, tracers_by_name["S"].data,
                                                              buoyancy.gravitational_acceleration, e.thermal_expansion,
                                                              e.haline_contraction, pHY.data))


def add_fplane_coriolis(grid, f, u, v, Gu, Gv, kernel_parameters=None):
    """- x_f_cross_U, - y_f_cross_U of the u, v tendencies for coriolis = FPlane(f)"""
    _lib.check(_lib.lib().ocn_add_fplane_coriolis(grid.handle, float(f), u.data, v.data, Gu.data, Gv.data, _range(kernel_parameters)))


def add_hydrostatic_pressure_gradient(grid, pHY, Gu, Gv, kernel_parameters=None):
    """-∂x pHY′, -∂y pHY′ of the u, v tendencies"""
    _lib.check(_lib.lib().ocn_add_hydrostatic_pressure_gradient(grid.handle, pHY.data, Gu.data, Gv.data, _range(kernel_parameters)))


def rk3_substep(grid, fields, Gn, Gm, Δt, γ, ζ):
    """rk3_substep_field! over a tuple of fields (ζ = None -> first stage)"""
    _lib.check(_lib.lib().ocn_rk3_substep(grid.handle, _ptr_array(fields), _ptr_array(Gn), _ptr_array(Gm),
                                          _loc_array(fields), len(fields), float(Δt), float(γ),
                                          0.0 if ζ is None else float(ζ), 0 if ζ is None else 1))


def cache_tendencies(grid, Gm, Gn):
    _lib.check(_lib.lib().ocn_cache_tendencies(grid.handle, _ptr_array(Gm), _ptr_array(Gn), _loc_array(Gn), len(Gn)))


def make_pressure_correction(grid, u, v, w, p):
    _lib.check(_lib.lib().ocn_make_pressure_correction(grid.handle, u.data, v.data, w.data, p.data))


def divide_interior(grid, p, divisor):
    _lib.check(_lib.lib().ocn_divide_interior(grid.handle, p.data, float(divisor)))
