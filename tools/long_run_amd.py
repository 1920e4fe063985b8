"""stability check of the configs[4] physics on the GPU box (stretched Bounded z, AMD, linear seawater, surface fluxes, evaporation):
many RK3 steps with the time-step wizard from a resting, stably stratified state with small noise"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oldoceananigans_jl_amd as ocn
from helpers import tanh_faces
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
arch = ocn.GPU(0)
Lz = 32.0
zf = Lz * (np.asarray(tanh_faces(N // 2)) - 0.0)           # faces in [-Lz, 0] if tanh_faces is [-1, 0]
grid = ocn.RectilinearGrid(arch, size=(N, N, N // 2), x=(0, 64.0), y=(0, 64.0), z=zf, topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
F = ocn.FieldBoundaryConditions
model = ocn.NonhydrostaticModel(grid=grid, tracers=("T", "S"), closure=ocn.AnisotropicMinimumDissipation(), coriolis=ocn.FPlane(f=1e-4),
                                buoyancy=ocn.SeawaterBuoyancy(ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4)),
                                boundary_conditions={"u": F(top=ocn.FluxBoundaryCondition(-1e-4)),
                                                     "T": F(top=ocn.FluxBoundaryCondition(5e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
                                                     "S": F(top=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(b=-1e-3 / 3600), field_dependencies="S"))})
rng = np.random.default_rng(0)
x, y, z = grid.nodes((ocn.Center, ocn.Center, ocn.Center))
T0 = 20 + 0.01 * z + 1e-4 * rng.standard_normal((N, N, N // 2)) * np.exp(z / 4.0)
ocn.set_model(model, u=1e-3 * rng.standard_normal((N, N, N // 2)), v=0.0, w=0.0, T=T0, S=35.0)
wizard = ocn.TimeStepWizard(cfl=0.5, max_change=1.1, max_Δt=20.0)
dt = 1.0
t0 = time.perf_counter()
for n in range(steps):
    if n % 50 == 0:
        dt = ocn.new_time_step(dt, wizard, model)
        u, T, S = model.velocities.u.interior(), model.tracers.T.interior(), model.tracers.S.interior()
        nu = model.diffusivity_fields[0].interior()
        print(f"step {n:4d} t = {model.clock.time:8.1f} s dt = {dt:6.2f} max|u| = {np.abs(u).max():.3e} T in [{T.min():.4f}, {T.max():.4f}] "
              f"<S> = {S.mean():.6f} max nu_e = {nu.max():.2e} div = {ocn.max_abs_divergence(model):.1e} nan = {ocn.hasnan(model)}", flush=True)
    ocn.time_step(model, dt)
print(f"{steps} steps in {time.perf_counter() - t0:.1f} s; nan = {ocn.hasnan(model)}")
