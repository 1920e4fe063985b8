"""Pressure solvers (reference: src/Solvers/{fft_based_poisson_solver,fourier_tridiagonal_poisson_solver,
batched_tridiagonal_solver}.jl). FFTs run on rocFFT through hipFFT; the batched tridiagonal solve along z is a
hand-written HIP kernel."""
import ctypes as C

import numpy as np

from . import _lib


class _PoissonSolver:
    kind = -1

    def __init__(self, grid):
        self.grid = grid
        h = C.c_void_p()
        _lib.check(_lib.lib().ocn_poisson_create(C.byref(h), grid.handle, self.kind))
        self.handle = h

    @property
    def architecture(self):
        return self.grid.architecture

    def set_source_term(self, rhs):
        """copy a host (Nx, Ny, Nz) real/complex array into the solver's complex right-hand-side storage"""
        a = np.asfortranarray(rhs, dtype=np.complex128)
        if a.shape != self.grid.size:
            raise ValueError(f"source term shape {a.shape} != {self.grid.size}")
        p = C.c_void_p()
        _lib.check(_lib.lib().ocn_poisson_rhs(self.handle, C.byref(p)))     # creates the complex storage on first use
        _lib.check(_lib.lib().ocn_memcpy_h2d(p, a.ctypes.data, a.nbytes))

    def close(self):
        if getattr(self, "handle", None) is not None:
            _lib.lib().ocn_poisson_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FFTBasedPoissonSolver(_PoissonSolver):
    """FFTBasedPoissonSolver(grid) (fft_based_poisson_solver.jl:52-74)"""
    kind = 0


class FourierTridiagonalPoissonSolver(_PoissonSolver):
    """FourierTridiagonalPoissonSolver(grid) with tridiagonal direction z (fourier_tridiagonal_poisson_solver.jl:75-134).
    Unlike the reference, `set_source_term` expects the rhs ALREADY multiplied by Δzᶜ (what
    `_fourier_tridiagonal_source_term!` produces, solve_for_pressure.jl:36-42)."""
    kind = 1


def solve(phi, solver):
    """solve!(ϕ, solver): consumes solver storage, writes interior(ϕ)."""
    _lib.check(_lib.lib().ocn_poisson_solve(solver.handle, phi.data))
    return phi


def solve_for_pressure(pressure, solver, velocities):
    """solve_for_pressure!(pressure, solver, Δt, Ũ) (solve_for_pressure.jl:91-95)"""
    u, v, w = velocities
    _lib.check(_lib.lib().ocn_solve_for_pressure(solver.handle, u.data, v.data, w.data, pressure.data))
    return pressure


def batched_tridiagonal_solve_z(a, b, c, f):
    """solve!(ϕ, BatchedTridiagonalSolver(grid; lower_diagonal=a, diagonal=b, upper_diagonal=c), f), z direction
    (batched_tridiagonal_solver.jl:110-133, 213-245). Host arrays in, host array out."""
    b = np.asfortranarray(b, dtype=np.float64)
    f = np.asfortranarray(f, dtype=np.complex128)
    Nx, Ny, Nz = b.shape
    a = np.ascontiguousarray(a, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    L = _lib.lib()
    ptrs = []

    def dev(arr, nbytes=None):
        p = C.c_void_p()
        _lib.check(L.ocn_malloc(C.byref(p), nbytes if nbytes is not None else max(arr.nbytes, 8)))
        if arr is not None and arr.nbytes:
            _lib.check(L.ocn_memcpy_h2d(p, arr.ctypes.data, arr.nbytes))
        ptrs.append(p)
        return p

    try:
        da, db, dc, df = dev(a), dev(b), dev(c), dev(f)
        dt, dphi = dev(None, b.nbytes), dev(None, f.nbytes)
        _lib.check(L.ocn_batched_tridiagonal_solve_z(Nx, Ny, Nz, da, db, dc, df, dt, dphi))
        out = np.empty_like(f)
        _lib.check(L.ocn_memcpy_d2h(out.ctypes.data, dphi, out.nbytes))
    finally:
        for p in ptrs:
            L.ocn_free(p)
    return out
