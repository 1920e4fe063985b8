"""oldoceananigans.jl_amd -- MI355X-native NonhydrostaticModel RK3 time-step (hot path of Oceananigans v0.100.5).

Host-side mirror of the reference's Architecture / Grid / Field / launch! surface over the C ABI of
libocn_mi355x.so (include/ocn_mi355x.h). Import as `import oldoceananigans_jl_amd as ocn`."""
from . import _lib
from ._lib import OcnError, build
from .advection import (Centered, FluxFormAdvection, UpwindBiased, WENO, adapt_advection_order, inflate_halo_size,
                        required_halo_size_x, required_halo_size_y, required_halo_size_z)
from .architectures import GPU, architecture, ndevices, own_stream, set_option, synchronize
from .boundary_conditions import (BoundaryCondition, FieldBoundaryConditions, FluxBoundaryCondition,
                                  GradientBoundaryCondition, LinearFieldFlux, OpenBoundaryCondition, ValueBoundaryCondition, compute_flux_bcs)
from .buoyancy import BuoyancyTracer, FPlane, LinearEquationOfState, SeawaterBuoyancy
from .checkpointer import set_from_checkpoint, write_checkpoint
from .closures import AnisotropicMinimumDissipation, ScalarDiffusivity
from .fields import (CenterField, Field, XFaceField, YFaceField, ZFaceField, fill_halo_regions, interior, set_)
from .grids import (Bounded, Center, Face, Flat, FullyConnected, LeftConnected, Periodic, RectilinearGrid, RightConnected,
                    with_halo)
from .models import NonhydrostaticModel, max_abs_divergence, set_model, time_step, update_state
from .solvers import (FFTBasedPoissonSolver, FourierTridiagonalPoissonSolver, batched_tridiagonal_solve_z, solve,
                      solve_for_pressure)
from .simulations import (CFL, AdvectiveCFL, DiffusiveCFL, Callback, IterationInterval, NaNChecker, Simulation, TimeInterval, TimeStepWizard, cell_advection_timescale,
                          cell_diffusion_timescale, default_nan_checker, hasnan, new_time_step, reset, run, stop_iteration_exceeded,
                          stop_time_exceeded, wall_time_limit_exceeded)
from . import kernels

__all__ = [n for n in dir() if not n.startswith("_")]
