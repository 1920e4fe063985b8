"""TEST-ONLY backend for oldoceananigans.jl_amd.distributed: the same backend protocol as DeviceBackend, implemented on
host numpy arrays with the CPU oracle's kernels and numpy FFTs. It lets the world_size-2 gloo tests run the PRODUCT's
distributed orchestration (partitioning, halo packing order, transposes, async interior/strip split, RK3 sequencing)
without a GPU. Nothing in the product imports this file."""
import ctypes as C

import numpy as np

from oracle import oracle as O

CODE = {"Periodic": 0, "Bounded": 1, "FullyConnected": 2}


class OracleLocalGrid:
    """rank-local grid adapter with the attributes the distributed driver reads"""

    def __init__(self, size, x, y, z, topology, halo):
        self.topology = topology
        codes = tuple(CODE[t.__name__] for t in topology)
        self.o = O.Grid(size, halo=halo, topology=codes, x=x, y=y, z=z)
        self.Nx, self.Ny, self.Nz = size
        self.Hx, self.Hy, self.Hz = halo
        self.size, self.halo_size = tuple(size), tuple(halo)
        self.x0, self.y0, self.z0 = x[0], y[0], z[0]
        self.Δxᶜᵃᵃ, self.Δyᵃᶜᵃ = self.o.dc[0][0], self.o.dc[1][0]
        self.Δzᵃᵃᶜ = self.o.dc[2]

    def total_size(self, loc):
        return self.o.parent_size(tuple(1 if l.__name__ == "Face" else 0 for l in loc))

    def interior_size(self, loc):
        return self.size

    def nodes(self, loc):
        out = []
        for d in range(3):
            delta = self.o.dc[d][0]
            origin = (self.x0, self.y0, self.z0)[d]
            face = loc[d].__name__ == "Face"
            arr = origin + delta * (np.arange(self.size[d]) + (0.0 if face else 0.5))
            shape = [1, 1, 1]
            shape[d] = self.size[d]
            out.append(arr.reshape(shape))
        return out


class HostField:
    def __init__(self, grid, loc):
        self.grid, self.loc = grid, loc
        self.codes = tuple(1 if l.__name__ == "Face" else 0 for l in loc)
        self.a = grid.o.zeros(self.codes)

    def parent(self):
        return self.a

    def set(self, value):
        H, N = self.grid.halo_size, self.grid.size
        view = self.a[H[0]:H[0] + N[0], H[1]:H[1] + N[1], H[2]:H[2] + N[2]]
        if callable(value):
            view[...] = value(*self.grid.nodes(self.loc))
        else:
            view[...] = value
        return self


class CpuBackend:
    def __init__(self, ctx, grid, ntracers, ocn):
        self.ctx, self.grid, self.ntracers = ctx, grid, ntracers
        torch = ctx.torch
        g = grid.local
        F, Cc = ocn.Face, ocn.Center
        locs = [(F, Cc, Cc), (Cc, F, Cc), (Cc, Cc, F)] + [(Cc, Cc, Cc)] * ntracers
        self.U = [HostField(g, l) for l in locs]
        self.Gn = [HostField(g, l) for l in locs]
        self.Gm = [HostField(g, l) for l in locs]
        self.p = HostField(g, (Cc, Cc, Cc))
        self.R, self.rank = ctx.world, ctx.rank
        n = g.Nx * g.Ny * g.Nz
        self.send = torch.zeros(2 * n, dtype=torch.float64)
        self.recv = torch.zeros(2 * n, dtype=torch.float64)
        Px, Py, Pz = g.total_size((Cc, Cc, Cc))
        self.slab = g.Hx * Py * Pz
        self.bufs = [torch.zeros(self.slab * len(locs), dtype=torch.float64) for _ in range(4)]
        self.zfield = np.zeros((g.Nx, g.Ny, g.Nz), dtype=np.complex128, order="F")
        # eigenvalues of the GLOBAL grid (reconstruct_global_grid, distributed_fft_based_poisson_solver.jl:92-136)
        Nxg = g.Nx * self.R
        self.lam = []
        for N, L in ((Nxg, grid.Lx_global), (g.Ny, g.o.L[1]), (g.Nz, g.o.L[2])):
            lam = np.zeros(N)
            O.lib().oro_poisson_eigenvalues(N, L, 0, lam.ctypes.data_as(C.POINTER(C.c_double)))
            self.lam.append(lam)

    # halos ------------------------------------------------------------------------------------------------------
    def fill_local_halos(self, fields, fill_open_bcs):
        for f in fields:
            self.grid.local.o.fill_halo_regions(f.a, f.codes, fill_open_bcs)

    def pack_x(self, fields, depth=None):
        g = self.grid.local
        H, N = g.Hx, g.Nx
        d = H if depth is None else depth
        slab = self.slab // H * d
        n = len(fields) * slab
        ws, es, wr, er = (b[:n] for b in self.bufs)
        for q, f in enumerate(fields):
            ws[q * slab:(q + 1) * slab] = self.ctx.torch.from_numpy(f.a[H:H + d].ravel(order="F").copy())
            es[q * slab:(q + 1) * slab] = self.ctx.torch.from_numpy(f.a[H + N - d:H + N].ravel(order="F").copy())
        return ws, es, wr, er

    def unpack_x(self, fields, depth=None):
        g = self.grid.local
        H, N = g.Hx, g.Nx
        d = H if depth is None else depth
        slab = self.slab // H * d
        wr, er = self.bufs[2], self.bufs[3]
        for q, f in enumerate(fields):
            shape = (d,) + f.a.shape[1:]
            f.a[H - d:H] = wr[q * slab:(q + 1) * slab].numpy().reshape(shape, order="F")
            f.a[N + H:N + H + d] = er[q * slab:(q + 1) * slab].numpy().reshape(shape, order="F")

    # kernels ----------------------------------------------------------------------------------------------------
    def rk3_substep(self, dt, γ, ζ):
        for U, Gn, Gm in zip(self.U, self.Gn, self.Gm):
            self.grid.local.o.rk3_substep(U.a, U.codes, dt, γ, ζ, Gn.a, Gm.a)

    def swap_tendencies(self):
        self.Gn, self.Gm = self.Gm, self.Gn

    def compute_tendencies(self, rng=None):
        o = self.grid.local.o
        u, v, w = (f.a for f in self.U[:3])
        for which, G in zip("uvw", self.Gn[:3]):
            o.compute_G(which, u, v, w, G.a, rng=rng if rng is not None else None)
        for t in range(self.ntracers):
            o.compute_G("c", u, v, w, self.Gn[3 + t].a, c=self.U[3 + t].a, rng=rng)

    def source_term(self):
        u, v, w = (f.a for f in self.U[:3])
        self.zfield[...] = self.grid.local.o.source_term(u, v, w, False)

    def _chunks(self, a):
        return a.view(np.float64)

    def poisson_forward_yz(self):
        g = self.grid.local
        R, Nyl = self.R, g.Ny // self.R
        z = np.fft.fft(np.fft.fft(self.zfield, axis=2), axis=1)
        out = np.empty((R, g.Nx, Nyl, g.Nz), dtype=np.complex128)
        for d in range(R):
            out[d] = z[:, d * Nyl:(d + 1) * Nyl, :]
        flat = np.concatenate([out[d].ravel(order="F") for d in range(R)])
        self.send[:] = self.ctx.torch.from_numpy(flat.view(np.float64).copy())

    def poisson_solve_x(self):
        g = self.grid.local
        R, Nyl, Nxg = self.R, g.Ny // self.R, g.Nx * self.R
        chunk = g.Nx * Nyl * g.Nz
        r = self.recv.numpy().view(np.complex128)
        x = np.empty((Nxg, Nyl, g.Nz), dtype=np.complex128)
        for s in range(R):
            x[s * g.Nx:(s + 1) * g.Nx] = r[s * chunk:(s + 1) * chunk].reshape((g.Nx, Nyl, g.Nz), order="F")
        x = np.fft.fft(x, axis=0)
        joff = self.rank * Nyl
        lam = (self.lam[0][:, None, None] + self.lam[1][None, joff:joff + Nyl, None]) + self.lam[2][None, None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            x = -x / lam
        if joff == 0:
            x[0, 0, 0] = 0
        x = np.fft.ifft(x, axis=0)
        flat = np.concatenate([x[s * g.Nx:(s + 1) * g.Nx].ravel(order="F") for s in range(R)])
        self.send[:] = self.ctx.torch.from_numpy(flat.view(np.float64).copy())

    def poisson_backward_yz(self):
        g = self.grid.local
        R, Nyl = self.R, g.Ny // self.R
        chunk = g.Nx * Nyl * g.Nz
        r = self.recv.numpy().view(np.complex128)
        z = np.empty((g.Nx, g.Ny, g.Nz), dtype=np.complex128)
        for d in range(R):
            z[:, d * Nyl:(d + 1) * Nyl, :] = r[d * chunk:(d + 1) * chunk].reshape((g.Nx, Nyl, g.Nz), order="F")
        z = np.fft.ifft(np.fft.ifft(z, axis=1), axis=2)
        H = g.halo_size
        self.p.a[H[0]:H[0] + g.Nx, H[1]:H[1] + g.Ny, H[2]:H[2] + g.Nz] = z.real

    def pressure_correction(self, rng=None):
        u, v, w = (f.a for f in self.U[:3])
        if rng is None:
            self.grid.local.o.pressure_correct(u, v, w, self.p.a)
            return
        # the oracle corrects whole fields: correct copies and keep the requested x range (test infrastructure, not the product)
        cu, cv, cw = u.copy(order="F"), v.copy(order="F"), w.copy(order="F")
        self.grid.local.o.pressure_correct(cu, cv, cw, self.p.a)
        H = self.grid.local.Hx
        sl = slice(H + rng[0] - 1, H + rng[1])
        for a, c in ((u, cu), (v, cv), (w, cw)):
            a[sl] = c[sl]

    def divide_pressure(self, divisor):
        g = self.grid.local
        H = g.halo_size
        self.p.a[H[0]:H[0] + g.Nx, H[1]:H[1] + g.Ny, H[2]:H[2] + g.Nz] /= divisor

    def synchronize(self):
        pass
