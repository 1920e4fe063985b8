"""Turbulence closures on the accelerated path: ScalarDiffusivity with constant isotropic ν, κ and explicit time
discretisation (reference: TurbulenceClosures/turbulence_closure_implementations/scalar_diffusivity.jl), and (8f.2)
AnisotropicMinimumDissipation with constant Poincaré coefficients (…/anisotropic_minimum_dissipation.jl)."""
import ctypes as C

import numpy as np


class ScalarDiffusivity:
    """ScalarDiffusivity(ν = 0, κ = 0): κ a number (all tracers) or a dict tracer-name -> number (scalar_diffusivity.jl:21-118)."""

    def __init__(self, ν=0.0, κ=0.0, nu=None, kappa=None):
        ν = ν if nu is None else nu
        κ = κ if kappa is None else kappa
        if callable(ν) or callable(κ) or (isinstance(κ, dict) and any(callable(x) for x in κ.values())):
            raise NotImplementedError("only constant (Number) viscosity / diffusivity is on the accelerated path")
        self.ν, self.κ = float(ν), ({n: float(v) for n, v in κ.items()} if isinstance(κ, dict) else float(κ))   # convert_diffusivity(FT, κ)
        if self.ν < 0:
            raise ValueError("viscosity must be non-negative")

    def kappa_array(self, tracer_names):
        if isinstance(self.κ, dict):
            missing = [n for n in tracer_names if n not in self.κ]
            if missing:
                raise ValueError(f"κ is missing tracers {missing}")     # with_tracers(), scalar_diffusivity.jl:151-160
            vals = [float(self.κ[n]) for n in tracer_names]
        else:
            vals = [float(self.κ)] * len(tracer_names)
        if any(v < 0 for v in vals):
            raise ValueError("diffusivity must be non-negative")
        arr = np.ascontiguousarray(vals if vals else [0.0], dtype=np.float64)
        return arr, arr.ctypes.data_as(C.POINTER(C.c_double))

    def __repr__(self):
        return f"ScalarDiffusivity{{ExplicitTimeDiscretization}}(ν={self.ν}, κ={self.κ})"


class AnisotropicMinimumDissipation:
    """AnisotropicMinimumDissipation(; C = 1/3, Cν = nothing, Cκ = nothing, Cb = nothing) (anisotropic_minimum_dissipation.jl:128-139):
    Cκ a number (all tracers) or a dict tracer-name -> number. The eddy viscosity / diffusivities are the model's
    `diffusivity_fields` (νₑ, κₑ)."""

    def __init__(self, C=1 / 3, Cν=None, Cκ=None, Cb=None, Cnu=None, Ckappa=None):
        Cν = Cν if Cnu is None else Cnu
        Cκ = Cκ if Ckappa is None else Ckappa
        if Cb is not None:
            raise NotImplementedError("the (unvalidated) buoyancy modification Cb is not on the accelerated path")
        self.Cν = C if Cν is None else Cν
        self.Cκ = C if Cκ is None else Cκ
        if callable(self.Cν) or callable(self.Cκ) or (isinstance(self.Cκ, dict) and any(callable(x) for x in self.Cκ.values())):
            raise NotImplementedError("only constant (Number) Poincaré coefficients are on the accelerated path")
        self.Cν = float(self.Cν)
        self.Cb = None

    def Ckappa_array(self, tracer_names):
        import ctypes
        if isinstance(self.Cκ, dict):
            missing = [n for n in tracer_names if n not in self.Cκ]
            if missing:
                raise ValueError(f"Cκ is missing tracers {missing}")     # tracer_diffusivities via with_tracers (:141-144)
            vals = [float(self.Cκ[n]) for n in tracer_names]
        else:
            vals = [float(self.Cκ)] * len(tracer_names)
        arr = np.ascontiguousarray(vals if vals else [0.0], dtype=np.float64)
        return arr, arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

    def __repr__(self):
        return f"AnisotropicMinimumDissipation{{ExplicitTimeDiscretization}}(Cν={self.Cν}, Cκ={self.Cκ}, Cb=nothing)"
