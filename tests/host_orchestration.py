"""TEST HARNESS -- not part of the product. The host-orchestrated form of the partitioned RK3 time-step (round 1's implementation): the
stage order of update_state! / compute_pressure_correction! / make_pressure_correction! written in Python against a small `backend`
protocol, collectives over `torch.distributed`. The product runs the same step INSIDE libocn_mi355x.so (oldoceananigans.jl_amd/distributed.py:
Distributed, LibraryDistributedModel); this module stays as

  * the harness of the world_size-2 gloo CPU test (tests/test_distributed_cpu.py plugs the oracle-based tests/cpu_backend.py into it),
  * a second implementation the GPU tests compare the library's partitioned model against (`DeviceBackend` drives the library's RAW
    kernels -- ocn_pack_x_halos, ocn_dist_poisson_*, ocn_compute_tendencies ... -- through the C ABI on virtual ranks that share one card),
  * the place where the stage order can be read next to the reference's (file:line in the docstrings).

All kernels and collectives of a rank are ordered on ONE HIP stream (torch's current stream, handed to the library with ocn_set_stream).
Everything the package's `distributed` module exports is re-exported here, so tests write `import host_orchestration as dist`."""
import ctypes as C
import os

import numpy as np

from oldoceananigans_jl_amd import _lib
from oldoceananigans_jl_amd.advection import WENO
from oldoceananigans_jl_amd.closures import AnisotropicMinimumDissipation
from oldoceananigans_jl_amd.distributed import *          # noqa: F401,F403
from oldoceananigans_jl_amd.distributed import Partition, _LocalView, _broadcast_bytes, _regular_coordinate  # noqa: F401
from oldoceananigans_jl_amd.fields import Field, _loc_array, _ptr_array
from oldoceananigans_jl_amd.grids import (Bounded, Center, Face, FullyConnected, LeftConnected, Periodic, RectilinearGrid,   # noqa: F401
                                          RightConnected)

RK3 = dict(γ1=8 / 15, γ2=5 / 12, γ3=3 / 4, ζ2=-17 / 60, ζ3=-5 / 12)   # runge_kutta_3.jl:69-74 (FT rationals)


class DistributedContext:
    """`Distributed(GPU(); partition = Partition(R))`: torch.distributed process group + the library bound to this rank's
    GPU and to torch's current stream."""

    def __init__(self, rank, world, device, torch, dist, arch):
        self.rank, self.world, self.device = rank, world, device
        self.torch, self.dist, self.arch = torch, dist, arch
        self.partition = Partition(world)
        self.west, self.east = self.partition.neighbours(rank)

    @property
    def partitioned(self):
        """does x carry rank boundaries (FullyConnected topology, halo exchange, strips)? True for world > 1"""
        return self.world > 1

    # -- collectives ---------------------------------------------------------------------------------------------
    def exchange_start(self, west_send, east_send, west_recv, east_recv):
        """MPI.Isend/Irecv! to both neighbours (halo_communication.jl:300,326). Order of the ops makes the pairing
        unambiguous even when both neighbours are the same rank (R = 2). Returns the pending requests: on RCCL the
        transfers run on the communicator's stream, concurrently with kernels launched afterwards."""
        d = self.dist
        ops = [d.P2POp(d.isend, west_send, self.west), d.P2POp(d.irecv, east_recv, self.east),
               d.P2POp(d.isend, east_send, self.east), d.P2POp(d.irecv, west_recv, self.west)]
        return d.batch_isend_irecv(ops)

    @staticmethod
    def exchange_wait(reqs):
        """MPI.Waitall (halo_communication.jl:164-165): later work on the stream waits for the transfers"""
        for req in reqs:
            req.wait()

    def exchange(self, west_send, east_send, west_recv, east_recv):
        self.exchange_wait(self.exchange_start(west_send, east_send, west_recv, east_recv))

    def all_to_all(self, recv, send):
        """MPI.Alltoallv! with equal counts (distributed_transpose.jl:185-191)"""
        self.dist.all_to_all_single(recv, send)

    def all_gather(self, gathered, payload):
        """MPI.Allgather of equal pieces: rank r's payload lands at gathered[r * n : (r + 1) * n] on every rank"""
        self.dist.all_gather_into_tensor(gathered, payload)

    def allreduce_max(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        self.dist.barrier()


class SelfLoopContext(DistributedContext):
    """MEASUREMENT ONLY: one rank that is its own west and east neighbour. The model runs the complete N > 1 code path --
    FullyConnected x, halo pack / exchange / unpack, interior + buffer strips, thin exchanges, substructured solve with an
    all-gather -- with device-to-device copies in place of the RCCL transfers, so the LOCAL cost of the partitioned path can
    be timed on a one-GPU box (bench.py: OCN_SELF_LOOP=1). The result equals the one-rank Periodic run."""

    partitioned = True

    def exchange_start(self, west_send, east_send, west_recv, east_recv):
        west_recv.copy_(east_send)        # what goes out of the east side comes in from the west (periodic ring of one)
        east_recv.copy_(west_send)
        return []

    def all_to_all(self, recv, send):
        if recv.data_ptr() != send.data_ptr():
            recv.copy_(send)

    def all_gather(self, gathered, payload):
        gathered.copy_(payload)

    def allreduce_max(self, value):
        return float(value)

    def barrier(self):
        pass


class HostStagedContext(DistributedContext):
    """REHEARSAL ONLY (never selected by default): the same ranks, buffers and call order as DistributedContext, but every
    collective is staged through host memory and run over gloo, so that N processes can share ONE card (RCCL refuses two
    ranks on one device). It exists to run the real multi-process DeviceBackend + bench.py path on a one-GPU box; its
    timings mean nothing."""

    def _host(self, t):
        self.torch.cuda.current_stream().synchronize()
        return t.cpu()

    def exchange_start(self, west_send, east_send, west_recv, east_recv):
        d = self.dist
        ws, es = self._host(west_send), self._host(east_send)
        wr, er = self.torch.empty_like(ws), self.torch.empty_like(es)
        ops = [d.P2POp(d.isend, ws, self.west, tag=1), d.P2POp(d.irecv, er, self.east, tag=1),
               d.P2POp(d.isend, es, self.east, tag=2), d.P2POp(d.irecv, wr, self.west, tag=2)]
        for req in d.batch_isend_irecv(ops):
            req.wait()
        west_recv.copy_(wr)
        east_recv.copy_(er)
        return []

    def all_to_all(self, recv, send):
        # gloo has no all_to_all_single: equal pieces, piece r of `send` goes to rank r
        d, R = self.dist, self.world
        h = self._host(send).reshape(R, -1)
        out = self.torch.empty_like(h)
        reqs = []
        for r in range(R):
            if r == self.rank:
                out[r].copy_(h[r])
            else:
                reqs.append(d.isend(h[r].contiguous(), r, tag=10 + self.rank))
        bufs = {}
        for r in range(R):
            if r != self.rank:
                bufs[r] = self.torch.empty_like(h[r])
                reqs.append(d.irecv(bufs[r], r, tag=10 + r))
        for req in reqs:
            req.wait()
        for r, b in bufs.items():
            out[r].copy_(b)
        recv.copy_(out.reshape(recv.shape))

    def all_gather(self, gathered, payload):
        h = self._host(payload)
        out = self.torch.empty(self.world * h.numel(), dtype=h.dtype)
        self.dist.all_gather_into_tensor(out, h.reshape(-1))
        gathered.copy_(out.reshape(gathered.shape))

    def allreduce_max(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def init_process_group(local_rank=0, backend=None, rehearse_on_one_gpu=False, self_loop=False):
    """one process per GPU; MASTER_ADDR/PORT, RANK, WORLD_SIZE come from torch.distributed.run.
    rehearse_on_one_gpu: all ranks on card 0, collectives staged through the host over gloo (HostStagedContext)"""
    if _lib._lib is not None and _lib.LOADED_BEFORE_TORCH:
        raise _lib.OcnError("libocn_mi355x.so was loaded before torch: torch bundles its own ROCm runtime under the same "
                            "sonames and cannot initialise on top of the system one. Import torch (or this module) and call "
                            "init_process_group() before creating any ocn.GPU().")
    # the host driver of this pool only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle: invalid argument
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from oldoceananigans_jl_amd.architectures import GPU
    use_gpu = torch.cuda.is_available()
    if rehearse_on_one_gpu:
        if not use_gpu:
            raise _lib.OcnError("rehearse_on_one_gpu needs a GPU")
        torch.cuda.set_device(0)
        if not dist.is_initialized():
            dist.init_process_group("gloo")
        arch = GPU(0)
        _lib.check(_lib.lib().ocn_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return HostStagedContext(dist.get_rank(), dist.get_world_size(), torch.device("cuda", 0), torch, dist, arch)
    if backend is None:
        backend = "nccl" if use_gpu else "gloo"
    if not dist.is_initialized():
        if use_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    if self_loop:
        if world != 1 or not use_gpu:
            raise _lib.OcnError("self_loop measures the partitioned path with ONE rank on a GPU")
        arch = GPU(local_rank)
        _lib.check(_lib.lib().ocn_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return SelfLoopContext(0, 1, torch.device("cuda", local_rank), torch, dist, arch)
    if use_gpu:
        arch = GPU(local_rank)
        # all library work goes to torch's current stream so RCCL ops are stream-ordered with the kernels
        _lib.check(_lib.lib().ocn_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        device = torch.device("cuda", local_rank)
    else:
        arch, device = None, torch.device("cpu")
    return DistributedContext(rank, world, device, torch, dist, arch)


# ----------------------------------------------------------------------------------------------------------------------
# grid
# ----------------------------------------------------------------------------------------------------------------------


# ----------------------------------------------------------------------------------------------------------------------
# product backend: HIP kernels through the C ABI, torch CUDA tensors as communication buffers
# ----------------------------------------------------------------------------------------------------------------------
class DeviceBackend:
    def __init__(self, ctx, grid, ntracers):
        if getattr(grid, "irregular", False) or getattr(grid, "global_x_topology", Periodic) is Bounded or getattr(grid, "partition", (1, 1))[1] > 1:
            raise NotImplementedError("irregular, Bounded and pencil partitions run through LibraryDistributedModel (gathered pressure solve)")
        self.ctx, self.grid, self.ntracers = ctx, grid, ntracers
        torch = ctx.torch
        g = grid.local
        L = _lib.lib()
        locs = [(Face, Center, Center), (Center, Face, Center), (Center, Center, Face)] + [(Center,) * 3] * ntracers
        self.U = [Field(l, g) for l in locs]
        self.U2 = [Field(l, g) for l in locs]      # target of the fused substeps (see time_step); swaps with U twice per step
        self.fused_substep = None                  # (Δt, γ, ζ) of the next stage while a fused evaluation is in flight
        self.Gn = [Field(l, g) for l in locs]
        self.Gm = [Field(l, g) for l in locs]
        self.p = Field((Center,) * 3, g)
        self.p2 = Field((Center,) * 3, g)       # receives p / Δt from the pressure-correction pass; swapped in as the pressure afterwards
        nf = len(locs)
        mk = lambda n: torch.zeros(n, dtype=torch.float64, device=ctx.device)   # noqa: E731
        total = sum(self._slab(f) for f in self.U)
        self.ws, self.es, self.wr, self.er = mk(total), mk(total), mk(total), mk(total)
        h = C.c_void_p()
        _lib.check(L.ocn_dist_poisson_create(C.byref(h), g.handle, ctx.world, ctx.rank, grid.Lx_global))
        self.solver = h
        n = C.c_size_t()
        _lib.check(L.ocn_dist_poisson_payload_size(h, C.byref(n)))
        if n.value:
            # substructured x solve: one small all-gather (2 values per mode) replaces the two all-to-alls
            self.payload = torch.zeros(2 * n.value, dtype=torch.float64, device=ctx.device)
            self.gathered = torch.zeros(2 * n.value * ctx.world, dtype=torch.float64, device=ctx.device)
            _lib.check(L.ocn_dist_poisson_set_gather_buffers(h, C.c_void_p(self.payload.data_ptr()), C.c_void_p(self.gathered.data_ptr())))
        else:
            _lib.check(L.ocn_dist_poisson_buffer_size(h, C.byref(n)))
            self.send = torch.zeros(2 * n.value, dtype=torch.float64, device=ctx.device)
            # one rank: the "transposes" are the identity -- alias the buffers instead of copying
            self.recv = self.send if ctx.world == 1 else torch.zeros(2 * n.value, dtype=torch.float64, device=ctx.device)
            _lib.check(L.ocn_dist_poisson_set_buffers(h, C.c_void_p(self.send.data_ptr()), C.c_void_p(self.recv.data_ptr())))
        self.profile, self.events, self.n_evals = False, [], 0

    # -- halos ---------------------------------------------------------------------------------------------------
    def set_boundary_conditions(self, bcs_by_index):
        """{index into U: FieldBoundaryConditions}: constant Flux / Value / Gradient / Open conditions on y / z sides"""
        self.bcs = dict(bcs_by_index)
        for fb in self.bcs.values():
            if any(s in fb.sides for s in ("west", "east")):
                raise NotImplementedError("the partitioned x direction is Periodic: no west / east conditions")

    def fill_local_halos(self, fields, fill_open_bcs):
        from oldoceananigans_jl_amd.fields import fill_halo_regions as fill
        bcs = getattr(self, "bcs", None)
        if not bcs:
            fill(fields, fill_open_bcs)
            return
        index = {id(f): n for n, f in enumerate(self.U)}
        fill(fields, fill_open_bcs, boundary_conditions=[bcs.get(index.get(id(f), -1)) for f in fields])

    def flux_bc_tendencies(self):
        """compute_flux_bc_tendencies! (compute_nonhydrostatic_tendencies.jl:170-184)"""
        from oldoceananigans_jl_amd.boundary_conditions import SIDES, compute_flux_bcs
        names = ["u", "v", "w"] + list(getattr(self, "tracer_names", ()))
        for n, fb in getattr(self, "bcs", {}).items():
            if any(bc.classification == "Flux" and bc.condition != 0.0 for bc in fb.sides.values()):
                compute_flux_bcs(self.Gn[n], fb)
        for n, fb in getattr(self, "bcs", {}).items():
            for side, bc in fb.sides.items():
                if getattr(bc, "linear", None) is not None:          # flux = a + b φ[i, j, k_boundary]
                    a, b_, dep = bc.linear
                    G = self.Gn[n]
                    loc = (C.c_int * 3)(*[1 if l is Face else 0 for l in G.loc])
                    _lib.check(_lib.lib().ocn_compute_linear_flux_bc(self.grid.local.handle, G.data, loc, SIDES.index(side), a, b_,
                                                                     self.U[names.index(dep)].data))

    def _slab(self, f):
        """doubles one field contributes per side: Hx x Py x Pz of ITS parent (Face fields on Bounded dims have one more plane)"""
        _, Py, Pz = self.grid.local.total_size(f.loc)
        return self.grid.local.Hx * Py * Pz

    def pack_x(self, fields, depth=None):
        """depth: columns per side (default Hx, the whole halo)"""
        Hx = self.grid.local.Hx
        depth = Hx if depth is None else int(depth)
        n = sum(self._slab(f) for f in fields) // Hx * depth
        _lib.check(_lib.lib().ocn_pack_x_halos_depth(self.grid.local.handle, _ptr_array(fields), _loc_array(fields), len(fields), depth,
                                                     C.c_void_p(self.ws.data_ptr()), C.c_void_p(self.es.data_ptr())))
        return self.ws[:n], self.es[:n], self.wr[:n], self.er[:n]

    def unpack_x(self, fields, depth=None):
        depth = self.grid.local.Hx if depth is None else int(depth)
        _lib.check(_lib.lib().ocn_unpack_x_halos_depth(self.grid.local.handle, _ptr_array(fields), _loc_array(fields), len(fields), depth,
                                                       C.c_void_p(self.wr.data_ptr()), C.c_void_p(self.er.data_ptr())))

    # -- kernels -------------------------------------------------------------------------------------------------
    def rk3_substep(self, dt, γ, ζ):
        from oldoceananigans_jl_amd import kernels
        kernels.rk3_substep(self.grid.local, self.U, self.Gn, self.Gm, dt, γ, ζ)

    def swap_tendencies(self):
        self.Gn, self.Gm = self.Gm, self.Gn

    def set_buoyancy(self, buoyancy, tracer_names):
        self.buoyancy, self.tracer_names = buoyancy, tuple(tracer_names)
        self.pHY = Field((Center,) * 3, self.grid.local)

    def update_hydrostatic_pressure(self):
        if getattr(self, "buoyancy", None) is not None:
            from oldoceananigans_jl_amd import kernels
            kernels.update_hydrostatic_pressure(self.grid.local, self.buoyancy, dict(zip(self.tracer_names, self.U[3:])), self.pHY)

    def can_fuse_substep(self):
        g = self.grid.local
        no_flux = not any(bc.classification == "Flux" and (bc.condition != 0.0 or getattr(bc, "linear", None) is not None)
                          for fb in getattr(self, "bcs", {}).values() for bc in fb.sides.values())
        return (no_flux and getattr(self, "closure", None) is None and getattr(self, "buoyancy", None) is None and
                getattr(self, "coriolis", None) is None and
                g.topology[1] is not Bounded and self.ntracers <= 3)

    def swap_prognostic(self):
        """after a fused evaluation: the updated fields become the live ones (list contents swap, Field objects stay)"""
        for a, b in zip(self.U, self.U2):
            a.data, b.data = b.data, a.data

    def compute_tendencies(self, rng=None):
        from oldoceananigans_jl_amd import kernels
        U = self.U
        ev = None
        if self.profile:
            torch = self.ctx.torch
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.fused_substep is None:
            kernels.compute_tendencies(self.grid.local, U[0], U[1], U[2], U[3:], self.Gn[0], self.Gn[1], self.Gn[2], self.Gn[3:],
                                       kernel_parameters=rng)
        else:
            Δt, γ, ζ = self.fused_substep
            kernels.compute_tendencies_and_substep(self.grid.local, U, self.Gn, self.U2, self.Gm, Δt, γ, ζ, kernel_parameters=rng)
        if ev:
            ev[1].record()
            self.events.append(ev)
        if getattr(self, "coriolis", None) is not None:
            kernels.add_fplane_coriolis(self.grid.local, self.coriolis.f, self.U[0], self.U[1], self.Gn[0], self.Gn[1], kernel_parameters=rng)
        if getattr(self, "buoyancy", None) is not None:
            kernels.add_hydrostatic_pressure_gradient(self.grid.local, self.pHY, self.Gn[0], self.Gn[1], kernel_parameters=rng)
        closure = getattr(self, "closure", None)
        if isinstance(closure, AnisotropicMinimumDissipation):
            kernels.compute_closure_tendencies_field(self.grid.local, self.U, self.Gn, self.nu_e, self.kappa_e, kernel_parameters=rng)
        elif closure is not None:
            kernels.compute_closure_tendencies(self.grid.local, self.U, self.Gn, closure, self.tracer_names, kernel_parameters=rng)

    def set_closure(self, closure, tracer_names):
        self.closure, self.tracer_names = closure, tuple(tracer_names)
        if isinstance(closure, AnisotropicMinimumDissipation):
            g = self.grid.local
            self.nu_e = Field((Center,) * 3, g)
            self.kappa_e = [Field((Center,) * 3, g) for _ in self.tracer_names]

    def compute_diffusivities(self):
        """compute_diffusivities! + fill_halo_regions!(diffusivity_fields; only_local_halos = true). The reference fills the x halos of
        νₑ, κₑ of a serial Periodic grid from the opposite side, i.e. with the closure evaluated there; an x-slab evaluates it at
        i = 0 and Nx + 1 itself from the exchanged velocity / tracer halos -- the same numbers, no extra exchange."""
        if not isinstance(getattr(self, "closure", None), AnisotropicMinimumDissipation):
            return
        from oldoceananigans_jl_amd import kernels
        from oldoceananigans_jl_amd.fields import fill_halo_regions as fill
        g = self.grid.local
        ext = 1 if self.ctx.partitioned else 0
        kernels.compute_amd_diffusivities(g, self.closure, self.tracer_names, self.U, self.nu_e, self.kappa_e,
                                          kernel_parameters=(1 - ext, g.Nx + ext, 1, g.Ny, 1, g.Nz))
        fill([self.nu_e] + self.kappa_e, True)

    def profile_read(self):
        """(total ms of the event-timed tendency launches, number of tendency EVALUATIONS) since the last read"""
        self.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.events)
        n = self.n_evals
        self.events, self.n_evals = [], 0
        return ms, n

    def source_term(self):
        U = self.U
        _lib.check(_lib.lib().ocn_dist_poisson_source_term(self.solver, U[0].data, U[1].data, U[2].data))

    def poisson_forward_local(self):
        _lib.check(_lib.lib().ocn_dist_poisson_forward_local(self.solver))

    def poisson_backward_local(self):
        _lib.check(_lib.lib().ocn_dist_poisson_backward_local(self.solver, self.p.data))

    def poisson_forward_yz(self):
        _lib.check(_lib.lib().ocn_dist_poisson_forward_yz(self.solver))

    def poisson_solve_x(self):
        _lib.check(_lib.lib().ocn_dist_poisson_solve_x(self.solver))

    def poisson_backward_yz(self):
        _lib.check(_lib.lib().ocn_dist_poisson_backward_yz(self.solver, self.p.data))

    def pressure_correction(self, rng=None):
        from oldoceananigans_jl_amd.kernels import _range
        U = self.U
        _lib.check(_lib.lib().ocn_make_pressure_correction_range(self.grid.local.handle, U[0].data, U[1].data, U[2].data, self.p.data,
                                                                 _range(rng)))

    def pressure_correction_divide(self, divisor, rng=None):
        """pressure correction over `rng` + p / divisor into the second pressure array (one pass instead of two)"""
        from oldoceananigans_jl_amd.kernels import _range
        U = self.U
        _lib.check(_lib.lib().ocn_make_pressure_correction_divide(self.grid.local.handle, U[0].data, U[1].data, U[2].data, self.p.data,
                                                                  self.p2.data, float(divisor), _range(rng)))

    def swap_pressure(self):
        self.p.data, self.p2.data = self.p2.data, self.p.data

    def divide_pressure(self, divisor):
        _lib.check(_lib.lib().ocn_divide_interior(self.grid.local.handle, self.p.data, float(divisor)))

    def max_abs_divergence(self):
        U, v = self.U, C.c_double()
        _lib.check(_lib.lib().ocn_max_abs_divergence(self.grid.local.handle, U[0].data, U[1].data, U[2].data, C.byref(v)))
        return v.value

    def synchronize(self):
        self.ctx.torch.cuda.current_stream().synchronize()

    def close(self):
        if getattr(self, "solver", None) is not None:
            _lib.lib().ocn_dist_poisson_destroy(self.solver)
            self.solver = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------------
# model + time stepping (backend-agnostic orchestration)
# ----------------------------------------------------------------------------------------------------------------------
class DistributedNonhydrostaticModel:
    """NonhydrostaticModel on a Distributed architecture (x-slabs): WENO(order=5), RK3, DistributedFFTBasedPoissonSolver."""

    def __init__(self, grid, advection=None, tracers=("T", "S"), timestepper="RungeKutta3", backend=None,
                 boundary_conditions=None, closure=None, buoyancy=None, coriolis=None):
        if advection is not None and not isinstance(advection, WENO):
            raise NotImplementedError("only advection = WENO(order=5) is on the accelerated hot path")
        self.grid, self.ctx = grid, grid.ctx
        self.tracer_names = tuple(tracers)
        self.backend = backend if backend is not None else DeviceBackend(grid.ctx, grid, len(self.tracer_names))
        if not hasattr(self.backend, "tracer_names"):
            self.backend.tracer_names = self.tracer_names
        self.time, self.iteration, self.stage = 0.0, 0, 1
        self.last_Δt = self.last_stage_Δt = float("inf")
        # overlap the halo exchange with the interior tendencies (AsynchronousDistributed)? None = automatic: only when the slab
        # is wide enough for whole-tile buffer strips (buffer_strip_width) -- on a thinner slab the two Hx-wide strips cost two
        # extra tile columns of the fused kernel, more than the exposed exchange; True / False force it
        self.async_halos = None
        if closure is not None:
            if hasattr(self.backend, "set_closure"):
                self.backend.set_closure(closure, self.tracer_names)
            else:
                self.backend.closure, self.backend.tracer_names = closure, self.tracer_names
        if coriolis is not None:
            self.backend.coriolis = coriolis
        if buoyancy is not None:
            missing = [t for t in buoyancy.required_tracers if t not in self.tracer_names]
            if missing:
                raise ValueError(f"{buoyancy!r} requires the tracers {missing}")
            self.backend.set_buoyancy(buoyancy, self.tracer_names)
        if boundary_conditions:
            names = ["u", "v", "w"] + list(self.tracer_names)
            unknown = [n for n in boundary_conditions if n not in names]
            if unknown:          # e.g. conditions on the diffusivity fields (νₑ, κₑ): the library-resident model carries them
                raise NotImplementedError(f"boundary conditions for {unknown}: use LibraryDistributedModel (the host-orchestrated model "
                                          "takes conditions on velocities and tracers only)")
            self.backend.set_boundary_conditions({names.index(n): fb for n, fb in boundary_conditions.items()})

    @property
    def clock(self):
        """model.clock (TimeSteppers/clock.jl:39-45): this model keeps time, iteration, stage, last_Δt, last_stage_Δt itself"""
        return self

    def reset(self):
        """reset!(model.clock) + zeroed tendencies (what ocn_model_reset does for the library's models)"""
        self.time, self.iteration, self.stage = 0.0, 0, 1
        self.last_Δt = self.last_stage_Δt = float("inf")
        import numpy as _np
        for f in list(self.backend.Gn) + list(self.backend.Gm):
            f.set_parent(_np.zeros(f.shape))

    # field access ---------------------------------------------------------------------------------------------
    def fields(self):
        names = ["u", "v", "w"] + list(self.tracer_names)
        return dict(zip(names, self.backend.U))

    @property
    def pressure(self):
        return self.backend.p

    def set_option(self, key, value):
        if key == "profile":
            self.backend.profile = bool(value)
            self.backend.events, self.backend.n_evals = [], 0
            return
        if key == "tendency_impl":
            if int(value) != 1:
                raise NotImplementedError("the distributed path always uses the fused tendency kernel")
            return
        from oldoceananigans_jl_amd.architectures import set_option
        set_option(key, value)

    def profile_read(self):
        return self.backend.profile_read()

    def fuse_substep_active(self):
        """whether time_step fuses rk3_substep! of stages 2 and 3 into the preceding tendency evaluation"""
        b = self.backend
        return bool(getattr(self, "fuse_substep", True) and hasattr(b, "can_fuse_substep") and b.can_fuse_substep())


def fill_halo_regions(model, fields, fill_open_bcs=True, x_fields=None, x_depth=None):
    """fill_halo_regions! of partitioned fields (halo_communication.jl:87-110): local boundary conditions first
    (boundary_condition_ordering.jl: DCBC last), then the x exchange.
    x_fields / x_depth: exchange only these fields' x halos, only this many columns deep (see compute_pressure_correction)"""
    b, ctx = model.backend, model.ctx
    b.fill_local_halos(fields, fill_open_bcs)
    if ctx.partitioned:
        xf = fields if x_fields is None else x_fields
        ws, es, wr, er = b.pack_x(xf, x_depth)
        ctx.exchange(ws, es, wr, er)
        b.unpack_x(xf, x_depth)


def solve_for_pressure(model):
    """solve_for_pressure! + solve!(::DistributedFFTBasedPoissonSolver) (distributed_fft_based_poisson_solver.jl:141-178)"""
    b, ctx = model.backend, model.ctx
    b.source_term()
    if getattr(b, "payload", None) is not None:
        # z Periodic: substructured solve along the partitioned direction -- local transforms and sweeps, one small all-gather
        b.poisson_forward_local()
        if ctx.partitioned:
            ctx.all_gather(b.gathered, b.payload)
        else:
            b.gathered.copy_(b.payload)
        b.poisson_backward_local()
        return
    b.poisson_forward_yz()
    if b.recv is not b.send:
        ctx.all_to_all(b.recv, b.send)    # transpose_y_to_x!
    b.poisson_solve_x()
    if b.recv is not b.send:
        ctx.all_to_all(b.recv, b.send)    # transpose_x_to_y!
    b.poisson_backward_yz()


def buffer_strip_width(model, Nx, Hx):
    """width of the two x strips that wait for the halos. The reference uses Hx (interleave_communication_and_computation.jl:
    69-119); any width >= Hx gives the same tendencies. The fused kernel works on 64-lane-wide tiles, so an Hx-wide strip costs
    as much as a 64-wide one (measured at 256^3: interior + two 3-wide strips 2.49 ms, interior + two 64-wide strips 1.51 ms,
    one launch 1.42 ms): strips are one whole tile wide whenever that leaves an interior of at least one tile."""
    W = getattr(model, "strip_width", None)
    if W is None:
        W = 64 if Nx >= 3 * 64 else Hx
    if not (Hx <= W and 2 * W < Nx):
        raise ValueError(f"strip width {W} must satisfy Hx <= W < Nx / 2")
    return W


def update_state(model, compute_tendencies=True):
    """update_state! (update_nonhydrostatic_model_state.jl:20-56) + compute_tendencies! with the interior / buffer split of
    interleave_communication_and_computation.jl:9-67 when halos are exchanged asynchronously."""
    b, ctx = model.backend, model.ctx
    g = model.grid.local
    if compute_tendencies and hasattr(b, "n_evals"):
        b.n_evals += 1
    reqs = getattr(model, "_halos_in_flight", None)
    if reqs is not None:
        # the x exchange was started by make_pressure_correction: finish the local fills (all columns are final now), take the
        # halos, then everything in one piece
        model._halos_in_flight = None
        b.fill_local_halos(b.U, False)
        ctx.exchange_wait(reqs)
        b.unpack_x(b.U)
        if hasattr(b, "compute_diffusivities"):
            b.compute_diffusivities()
        if hasattr(b, "update_hydrostatic_pressure"):
            b.update_hydrostatic_pressure()
        if compute_tendencies:
            b.compute_tendencies(None)
            if hasattr(b, "flux_bc_tendencies"):
                b.flux_bc_tendencies()
        return
    # with buoyancy, pHY′ in the x-halo columns needs the exchanged tracers: fill, integrate, then evaluate (no overlap)
    overlap = model.async_halos if model.async_halos is not None else g.Nx >= 3 * 64
    # eddy diffusivities and pHY′ in the x-halo columns need the exchanged fields: fill, evaluate them, then the tendencies (no overlap)
    if (not compute_tendencies or not ctx.partitioned or not overlap or g.Nx <= 2 * g.Hx or
            getattr(b, "buoyancy", None) is not None or isinstance(getattr(b, "closure", None), AnisotropicMinimumDissipation)):
        fill_halo_regions(model, b.U, fill_open_bcs=False)
        if hasattr(b, "compute_diffusivities"):
            b.compute_diffusivities()                     # compute_auxiliaries! (update_nonhydrostatic_model_state.jl:58-69)
        if hasattr(b, "update_hydrostatic_pressure"):
            b.update_hydrostatic_pressure()           # compute_auxiliaries! (update_nonhydrostatic_model_state.jl:58-69)
        if compute_tendencies:
            b.compute_tendencies(None)
            if hasattr(b, "flux_bc_tendencies"):
                b.flux_bc_tendencies()
        return
    # async: start the exchange, compute the interior that does not depend on x halos, finish, compute the two strips
    b.fill_local_halos(b.U, False)
    ws, es, wr, er = b.pack_x(b.U)
    Nx, Ny, Nz, Hx = g.Nx, g.Ny, g.Nz, g.Hx
    W = buffer_strip_width(model, Nx, Hx)
    reqs = ctx.exchange_start(ws, es, wr, er)                         # halos fly ...
    b.compute_tendencies((W + 1, Nx - W, 1, Ny, 1, Nz))              # ... while the interior is computed (:27-67)
    ctx.exchange_wait(reqs)                                           # synchronize_communication! (distributed_fields.jl:71-88)
    b.unpack_x(b.U)                                                   # complete_communication_and_compute_buffer! (:9-20)
    b.compute_tendencies((1, W, 1, Ny, 1, Nz))                        # compute_buffer_tendencies! west strip
    b.compute_tendencies((Nx - W + 1, Nx, 1, Ny, 1, Nz))              # east strip
    if hasattr(b, "flux_bc_tendencies"):
        b.flux_bc_tendencies()


def compute_pressure_correction(model):
    """compute_pressure_correction! (pressure_correction.jl:8-20)"""
    b = model.backend
    # Of the x halos only ONE column is read before update_state! fills everything again: u[Nx+1] by the divergence, p[0] by
    # the correction. The reference's generic fills move Hx columns of u, v, w and of p here; `thin_halos` (default) exchanges the
    # one column of u and of p -- 1/9 and 1/3 of the bytes, both exposed on the critical path -- with identical results in every
    # cell that is read.
    thin = getattr(model, "thin_halos", True)
    fill_halo_regions(model, b.U[:3], fill_open_bcs=True, x_fields=b.U[:1] if thin else None, x_depth=1 if thin else None)
    solve_for_pressure(model)
    fill_halo_regions(model, [b.p], fill_open_bcs=True, x_depth=1 if thin else None)


def make_pressure_correction(model, Δt, start_halo_exchange=False):
    """make_pressure_correction! (pressure_correction.jl:40-53).
    start_halo_exchange (the caller evaluates tendencies next): correct the two Hx-wide boundary strips first, fill their y / z halos,
    pack them and START the x exchange of the coming update_state!; the interior correction, p / Δt and the final local fills
    then run while the halos are in flight, and the tendencies need no interior / buffer split -- one full launch of the fused
    kernel (measured on one rank looped onto itself, 256^3: 1.92 ms for interior + two strips -> 1.73 ms). Same values everywhere:
    every cell is corrected once, the packed columns are final when they are packed."""
    b, ctx = model.backend, model.ctx
    g = model.grid.local
    dtp = max(np.finfo(np.float64).eps, Δt)
    fused = hasattr(b, "pressure_correction_divide")     # p / Δt written by the correction pass itself (second array, swapped in)
    if not (start_halo_exchange and ctx.partitioned and getattr(model, "early_exchange", True) and model.async_halos is not False and
            g.Nx > 2 * g.Hx and hasattr(b, "pack_x")):
        if fused:
            b.pressure_correction_divide(dtp)
            b.swap_pressure()
        else:
            b.pressure_correction()
            b.divide_pressure(dtp)
        return
    Nx, Ny, Nz, Hx = g.Nx, g.Ny, g.Nz, g.Hx
    pc = (lambda r: b.pressure_correction_divide(dtp, r)) if fused else b.pressure_correction
    pc((1, Hx, 1, Ny, 1, Nz))
    pc((Nx - Hx + 1, Nx, 1, Ny, 1, Nz))
    b.fill_local_halos(b.U, False)                       # the strips' y / z halos (corners ride along in the buffers)
    ws, es, wr, er = b.pack_x(b.U)
    model._halos_in_flight = ctx.exchange_start(ws, es, wr, er)
    pc((Hx + 1, Nx - Hx, 1, Ny, 1, Nz))
    if fused:
        b.swap_pressure()
    else:
        b.divide_pressure(dtp)


def set_model(model, enforce_incompressibility=True, **kwargs):
    """set!(model; kwargs...) with LOCAL interior arrays / functions of the local nodes (set_nonhydrostatic_model.jl:33-60)"""
    flds = model.fields()
    for name, value in kwargs.items():
        if name not in flds:
            raise ValueError(f"name {name} not found in model.velocities or model.tracers.")
        flds[name].set(value)
    b = model.backend
    fill_halo_regions(model, b.U, fill_open_bcs=True)
    update_state(model, compute_tendencies=False)
    if enforce_incompressibility:
        compute_pressure_correction(model)
        make_pressure_correction(model, 1.0)
        update_state(model, compute_tendencies=False)


def _tick(model, Δt, stage):
    model.time += Δt
    if stage:
        model.stage += 1
        model.last_stage_Δt = Δt
    else:
        model.iteration += 1
        model.stage = 1
        model.last_Δt = model.last_stage_Δt = Δt


def time_step(model, Δt):
    """time_step!(model, Δt) for RungeKutta3 (runge_kutta_3.jl:93-170) on the partitioned model"""
    b = model.backend
    if model.iteration == 0:
        update_state(model, True)
    γ = (RK3["γ1"], RK3["γ2"], RK3["γ3"])
    ζ = (None, RK3["ζ2"], RK3["ζ3"])
    stage_dt = (Δt * γ[0], Δt * (γ[1] + ζ[1]), Δt * (γ[2] + ζ[2]))
    tn1 = model.time + Δt
    # stages 2 and 3: rk3_substep! is fused into the tendency evaluation that precedes it (backend permitting)
    fuse = model.fuse_substep_active()
    substep_done = False
    for s in range(3):
        if not substep_done:
            b.rk3_substep(Δt, γ[s], ζ[s])
        substep_done = False
        if s < 2:
            _tick(model, stage_dt[s], True)
        else:
            corrected = tn1 - model.time
            _tick(model, stage_dt[2], False)
            model.last_stage_Δt, model.last_Δt = corrected, Δt
        compute_pressure_correction(model)
        make_pressure_correction(model, stage_dt[s], start_halo_exchange=True)      # update_state!(…; compute_tendencies) follows
        if s < 2:
            b.swap_tendencies()           # cache_previous_tendencies! as a pointer swap (see ocn_api.hip)
        if s < 2 and fuse:
            b.fused_substep = (Δt, γ[s + 1], ζ[s + 1])
            update_state(model, True)
            b.fused_substep = None
            b.swap_prognostic()
            substep_done = True
        else:
            update_state(model, True)


def max_abs_divergence(model):
    """global max |div u| (test helper)"""
    b, ctx = model.backend, model.ctx
    fill_halo_regions(model, b.U[:3], fill_open_bcs=True)
    if hasattr(b, "max_abs_divergence"):
        local = b.max_abs_divergence()
    else:
        g = model.grid.local
        u, v, w = (f.parent() for f in b.U[:3])
        H = g.Hx
        core = (slice(H, -H),) * 3
        dx, dy, dz = g.Δxᶜᵃᵃ, g.Δyᵃᶜᵃ, g.Δzᵃᵃᶜ[0]
        div = ((u[H + 1:u.shape[0] - H + 1, H:-H, H:-H] - u[core]) / dx + (v[H:-H, H + 1:v.shape[1] - H + 1, H:-H] - v[core]) / dy +
               (w[H:-H, H:-H, H + 1:w.shape[2] - H + 1] - w[core]) / dz)
        local = float(np.abs(div).max())
    return ctx.allreduce_max(local)