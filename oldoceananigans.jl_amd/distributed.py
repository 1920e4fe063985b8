"""Distributed x-slab NonhydrostaticModel (reference: src/DistributedComputations/, src/Models/interleave_communication_and_computation.jl).

MI355X-first design: ONE process per GPU. The PRODUCT path keeps everything behind the C ABI: the library owns the RCCL
communicator and runs the partitioned time-step itself (`Distributed`, `LibraryDistributedModel` at the end of this module ->
ocn_dist_create / ocn_dist_model_create / ocn_model_time_step; no torch in the process). The classes in between are the
host-orchestrated form of the same step, written against a small `backend` protocol: it is the TEST HARNESS (gloo world_size-2 CPU
tests, virtual ranks on one card, host-staged rehearsals) and the place where the stage order can be read next to the reference's.
There collectives go through `torch.distributed` and all kernels and collectives are ordered on ONE HIP stream per rank (torch's current
stream, handed to the library with `ocn_set_stream`), so there is no host synchronisation on the hot path -- the
reference calls `sync_device!` before every MPI call (halo_communication.jl:181, distributed_transpose.jl:187).

  * halo exchange (fill_halo_regions! of a partitioned field, halo_communication.jl:87-110,170-187): local y / z fills,
    then ONE batched send/recv pair with the two ring neighbours carrying all fields of the call
    (Hx x Py x Pz x nfields doubles per side -- corners ride along, communication_buffers.jl:53,71-76);
  * distributed FFT pressure solve (distributed_fft_based_poisson_solver.jl:141-188): local (y, z) FFT -> all-to-all
    (transpose y -> x) -> x FFT, spectral divide, inverse x FFT -> all-to-all (x -> y) -> inverse (y, z) FFT;
  * the time-step orchestration below is written against a small `backend` protocol: the product backend
    (`DeviceBackend`) drives the HIP kernels through the C ABI; the world_size-2 CPU tests plug a test-only backend
    (tests/cpu_backend.py, built on the oracle) into the SAME orchestration and run it over gloo.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .advection import WENO
from .closures import AnisotropicMinimumDissipation
from .fields import Field, _loc_array, _ptr_array
from .grids import (Bounded, Center, Face, FullyConnected, LeftConnected, Periodic, RectilinearGrid, RightConnected,
                    _regular_coordinate)

RK3 = dict(γ1=8 / 15, γ2=5 / 12, γ3=3 / 4, ζ2=-17 / 60, ζ3=-5 / 12)   # runge_kutta_3.jl:69-74 (FT rationals)


# ----------------------------------------------------------------------------------------------------------------------
# architecture / communicator
# ----------------------------------------------------------------------------------------------------------------------
class Partition:
    """Partition(Rx) (distributed_architectures.jl:14-63): x-slabs. Rank r has local index r (k fastest, :354) and ring
    neighbours with periodic wrap (:391-434)."""

    def __init__(self, R):
        self.R = int(R)

    def neighbours(self, rank, periodic=True):
        west, east = rank - 1, rank + 1
        if periodic:
            return west % self.R, east % self.R
        return (west if west >= 0 else None), (east if east < self.R else None)


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
    from .architectures import GPU
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
def local_sizes(N, R):
    """local_size(N, R, local_index) for every rank (distributed_grids.jl:44-58): N ÷ R cells per rank, the remainder on the last"""
    nl = [N // R] * R
    nl[-1] += N - sum(nl)
    return nl


def partition_coordinate(c, n_local, R, r):
    """partition_coordinate(c::Tuple, n, arch, dim) (partition_assemble.jl:63-76): Δl = (c₂ - c₁) / N, the intervals are chained
    l[i] = (l[i-1][2], l[i-1][2] + Δl nl[i]). `n_local`: one size (equal slabs) or the list of all ranks' sizes."""
    nl = [n_local] * R if isinstance(n_local, int) else list(n_local)
    N = sum(nl)
    dl = (c[1] - c[0]) / N
    lo, hi = c[0], c[0] + dl * nl[0]
    for i in range(1, r + 1):
        lo = hi
        hi = lo + dl * nl[i]
    return (lo, hi)


class DistributedRectilinearGrid:
    """RectilinearGrid(arch::Distributed; size = GLOBAL size, ...) (distributed_grids.jl:75-117): the rank-local grid of
    an x-slab partition. `local` is the RectilinearGrid-like object the kernels run on."""

    def __init__(self, ctx, size, x=None, y=None, z=None, extent=None, topology=(Periodic, Periodic, Periodic),
                 halo=(3, 3, 3), make_local_grid=None, partition=None):
        self.ctx = ctx
        # Partition(Rx, Ry) (distributed_architectures.jl:14-63): rank = ix * Ry + iy (index2rank, :354-389). Default: x-slabs.
        Rx, Ry = (ctx.world, 1) if partition is None else (int(partition[0]), int(partition[1]))
        if Rx * Ry != ctx.world:
            raise ValueError(f"Partition({Rx}, {Ry}) needs {Rx * Ry} ranks, the architecture has {ctx.world}")
        self.partition = (Rx, Ry)
        R, r = Rx, ctx.rank // Ry
        iy = ctx.rank % Ry
        if extent is not None:
            x, y, z = (0.0, float(extent[0])), (0.0, float(extent[1])), (-float(extent[2]), 0.0)
        self.global_size = tuple(int(n) for n in size)
        if topology[0] not in (Periodic, Bounded):
            raise ValueError("the partitioned direction is Periodic or Bounded")
        self.global_x_topology = topology[0]
        self._global_arguments = dict(size=self.global_size, x=x, y=y, z=z, topology=tuple(topology), halo=halo)
        # local_size (distributed_grids.jl:44-58): the remainder of Nx / R goes to the last rank
        self.local_sizes = local_sizes(self.global_size[0], R)
        self.irregular = len(set(self.local_sizes)) > 1
        nxl = self.local_sizes[r]
        self.x_global = (float(x[0]), float(x[1]))
        self.Lx_global = _regular_coordinate(x, self.global_size[0], "x")[1]
        xl = x if R == 1 else partition_coordinate(x, self.local_sizes, R, r)
        # insert_connected_topology (distributed_grids.jl:339-346)
        if not ctx.partitioned or (R == 1 and Ry > 1):     # y-slabs only: x stays what it is
            tx = topology[0]
        elif topology[0] is Periodic:
            tx = FullyConnected
        else:
            tx = RightConnected if r == 0 else (LeftConnected if r == R - 1 else FullyConnected)
        # the y direction of a pencil partition: connected local grids (insert_connected_topology again), the remainder on the last row
        self.local_sizes_y = local_sizes(self.global_size[1], Ry)
        self.Ly_global = _regular_coordinate(y, self.global_size[1], "y")[1]
        self.global_y_topology = topology[1]
        nyl = self.local_sizes_y[iy]
        ty = topology[1]
        if Ry > 1:
            if topology[1] not in (Periodic, Bounded):
                raise ValueError("a partitioned direction is Periodic or Bounded")
            if topology[1] is Periodic:
                ty = FullyConnected
            else:
                ty = RightConnected if iy == 0 else (LeftConnected if iy == Ry - 1 else FullyConnected)
            y = partition_coordinate(y, self.local_sizes_y, Ry, iy)
        self.j_offset = sum(self.local_sizes_y[:iy])
        topo = (tx, ty, topology[2])
        self.local_size = (nxl, nyl, self.global_size[2])
        self.i_offset = sum(self.local_sizes[:r])    # global index of local i = 1 minus one
        if make_local_grid is None:
            self.local = RectilinearGrid(ctx.arch, self.local_size, x=xl, y=y, z=z, topology=topo, halo=halo)
        else:
            self.local = make_local_grid(self.local_size, xl, y, z, topo, halo)

    def global_nodes(self, loc):
        """the node coordinates of the GLOBAL grid (reconstruct_global_grid, distributed_grids.jl:192-233) restricted to this rank's
        interior cells. The local grid's own nodes are ranges over the LOCAL interval (like the reference's) and differ from these by
        round-off; a state evaluated here is bit for bit the slab of the state a serial model evaluates on the global grid."""
        if getattr(self, "_global_grid", None) is None:
            a = self._global_arguments
            self._global_grid = RectilinearGrid(None, a["size"], x=a["x"], y=a["y"], z=a["z"], topology=a["topology"], halo=a["halo"])
        X, Y, Z = self._global_grid.nodes(loc)
        nx, ny, _ = self.local.interior_size(loc)
        return X[self.i_offset:self.i_offset + nx], Y[:, self.j_offset:self.j_offset + ny], Z

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.local, name)


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
        from .fields import fill_halo_regions as fill
        bcs = getattr(self, "bcs", None)
        if not bcs:
            fill(fields, fill_open_bcs)
            return
        index = {id(f): n for n, f in enumerate(self.U)}
        fill(fields, fill_open_bcs, boundary_conditions=[bcs.get(index.get(id(f), -1)) for f in fields])

    def flux_bc_tendencies(self):
        """compute_flux_bc_tendencies! (compute_nonhydrostatic_tendencies.jl:170-184)"""
        from .boundary_conditions import SIDES, compute_flux_bcs
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
        from . import kernels
        kernels.rk3_substep(self.grid.local, self.U, self.Gn, self.Gm, dt, γ, ζ)

    def swap_tendencies(self):
        self.Gn, self.Gm = self.Gm, self.Gn

    def set_buoyancy(self, buoyancy, tracer_names):
        self.buoyancy, self.tracer_names = buoyancy, tuple(tracer_names)
        self.pHY = Field((Center,) * 3, self.grid.local)

    def update_hydrostatic_pressure(self):
        if getattr(self, "buoyancy", None) is not None:
            from . import kernels
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
        from . import kernels
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
        from . import kernels
        from .fields import fill_halo_regions as fill
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
        from .kernels import _range
        U = self.U
        _lib.check(_lib.lib().ocn_make_pressure_correction_range(self.grid.local.handle, U[0].data, U[1].data, U[2].data, self.p.data,
                                                                 _range(rng)))

    def pressure_correction_divide(self, divisor, rng=None):
        """pressure correction over `rng` + p / divisor into the second pressure array (one pass instead of two)"""
        from .kernels import _range
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
        from .architectures import set_option
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


def local_initial_state(model, fn):
    """evaluate a global initial-state generator on this rank's slab: `fn(ocn, model_like)` is called with an object whose
    grid.nodes() are the LOCAL node coordinates"""
    import sys
    return fn(sys.modules[__name__.rsplit(".", 1)[0]], _LocalView(model))


class _LocalView:
    def __init__(self, model):
        self.grid = model.grid.local
        self._model = model

    def fields(self):
        return self._model.fields()


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


# ----------------------------------------------------------------------------------------------------------------------
# PRODUCT PATH: the communicator and the partitioned time-step inside libocn_mi355x.so (include/ocn_mi355x.h, distributed group)
# ----------------------------------------------------------------------------------------------------------------------
class Distributed:
    """`Distributed(GPU(); partition = Partition(R))` (distributed_architectures.jl:166-302) with the communicator INSIDE the library:
    RCCL (ocn_dist_create) or caller-supplied collectives (ocn_dist_create_transport). No torch in the process."""

    def __init__(self, handle, rank, world, arch, self_loop=False, keepalive=None):
        self.handle, self.rank, self.world, self.arch = handle, rank, world, arch
        self.partition = Partition(world)
        self.west, self.east = self.partition.neighbours(rank)
        self.self_loop = bool(self_loop)
        self._keepalive = keepalive            # ctypes callbacks of a transport must outlive the communicator

    @property
    def partitioned(self):
        return self.world > 1 or self.self_loop

    def allreduce_max(self, value):
        v = C.c_double(float(value))
        _lib.check(_lib.lib().ocn_dist_allreduce_max(self.handle, C.byref(v)))
        return v.value

    def barrier(self):
        _lib.check(_lib.lib().ocn_dist_barrier(self.handle))

    def info(self):
        """what the communicator itself reports (ocn_dist_info + ocn_dist_comm_info): for the RCCL transport the number of ranks and
        this rank's index come from ncclCommCount / ncclCommUserRank -- the ranks RCCL really connected"""
        w, r, we, ea = (C.c_int() for _ in range(4))
        _lib.check(_lib.lib().ocn_dist_info(self.handle, C.byref(w), C.byref(r), C.byref(we), C.byref(ea)))
        k, n, cr, dev = (C.c_int() for _ in range(4))
        _lib.check(_lib.lib().ocn_dist_comm_info(self.handle, C.byref(k), C.byref(n), C.byref(cr), C.byref(dev)))
        return {"world": w.value, "rank": r.value, "west": we.value, "east": ea.value,
                "transport": "rccl" if k.value == 0 else "caller-supplied collectives (ocn_transport_t)",
                "comm_ranks": n.value, "comm_rank": cr.value, "device": dev.value, "self_loop": self.self_loop}

    def close(self):
        if getattr(self, "handle", None) is not None:
            _lib.lib().ocn_dist_destroy(self.handle)
            self.handle = None

    # -- construction --------------------------------------------------------------------------------------------
    @classmethod
    def rccl(cls, arch, unique_id, world, rank, self_loop=False):
        h = C.c_void_p()
        with _stdout_to_stderr():        # RCCL prints a version / host banner on stdout when it initialises
            _lib.check(_lib.lib().ocn_dist_create(C.byref(h), unique_id, int(world), int(rank)))
        if self_loop:
            _lib.check(_lib.lib().ocn_dist_set_self_loop(h, 1))
        return cls(h, rank, world, arch, self_loop)

    @classmethod
    def from_environment(cls, local_rank=None, self_loop=False):
        """one process per GPU under `python -m torch.distributed.run` (or any launcher that exports RANK, WORLD_SIZE, LOCAL_RANK,
        MASTER_ADDR, MASTER_PORT): rank 0 makes the ncclUniqueId, a one-shot TCP exchange carries it to the other ranks"""
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # this pool's driver only supports dmabuf IPC
        from .architectures import GPU
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if local_rank is None:
            local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # device!(child_architecture, node_rank % ndevices(child_architecture)) (distributed_architectures.jl:284-288): also right when the
        # launcher shows every process one card only
        from .architectures import ndevices
        arch = GPU(local_rank % max(1, ndevices()))
        uid = C.create_string_buffer(128)
        if rank == 0:
            with _stdout_to_stderr():
                _lib.check(_lib.lib().ocn_dist_unique_id(uid))
        if world > 1:
            uid = C.create_string_buffer(_broadcast_bytes(uid.raw if rank == 0 else None, rank, world), 128)
        return cls.rccl(arch, uid, world, rank, self_loop)

    @classmethod
    def transport(cls, arch, collectives, world, rank, self_loop=False):
        """collectives: an object with exchange_start(west_send, east_send, west_recv, east_recv, count), exchange_wait(),
        all_to_all(send, recv, count_per_rank), all_gather(send, recv, count), allreduce_max(value) -> float; buffer arguments are
        device addresses (int). This is the seam an MPI binder would use; the tests plug in threads sharing one card."""
        T = _lib.Transport

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return 0
                except Exception:          # a Python exception must not unwind through C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            return wrapped

        def allreduce(user, pvalue):
            pvalue[0] = float(collectives.allreduce_max(pvalue[0]))

        cbs = dict(
            exchange_start=T.EXCHANGE_START(guard(lambda u, ws, es, wr, er, n, st: collectives.exchange_start(ws, es, wr, er, n))),
            exchange_wait=T.EXCHANGE_WAIT(guard(lambda u, st: collectives.exchange_wait())),
            all_to_all=T.ALL_TO_ALL(guard(lambda u, s_, r_, n, st: collectives.all_to_all(s_, r_, n))),
            all_gather=T.ALL_GATHER(guard(lambda u, s_, r_, n, st: collectives.all_gather(s_, r_, n))),
            allreduce_max=T.ALLREDUCE_MAX(guard(allreduce)))
        if hasattr(collectives, "exchange_peers"):      # pencil partitions: exchange with an explicit pair of peers
            cbs["exchange_peers"] = T.EXCHANGE_PEERS(guard(lambda u, pl, ph, ls, hs, lr, hr, n, st: collectives.exchange_peers(pl, ph, ls, hs, lr, hr, n)))
        t = T(user=None, **cbs)
        h = C.c_void_p()
        _lib.check(_lib.lib().ocn_dist_create_transport(C.byref(h), C.byref(t), int(world), int(rank)))
        if self_loop:
            _lib.check(_lib.lib().ocn_dist_set_self_loop(h, 1))
        return cls(h, rank, world, arch, self_loop, keepalive=(t, cbs, collectives))


_BOOT_MAGIC = b"OCN-RCCL-ID-1"


import contextlib


@contextlib.contextmanager
def _stdout_to_stderr():
    """file descriptor 1 points at stderr inside the block: what a C library prints on stdout (RCCL's start-up banner) must not land in
    the one-JSON-line output of bench.py"""
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        os.dup2(saved, 1)
        os.close(saved)


def _broadcast_bytes(payload, rank, world, timeout=120.0):
    """rank 0 -> everyone, one shot over TCP on MASTER_ADDR, first free port of MASTER_PORT + 1 .. + 16 (the launcher's own store sits
    on MASTER_PORT); a magic prefix tells our listener from a stranger's"""
    import socket
    import time
    addr, base = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500"))
    ports = [base + 1 + k for k in range(16)]
    if rank == 0:
        srv = None
        for port in ports:
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, port))
                break
            except OSError:
                srv.close()
                srv = None
        if srv is None:
            raise _lib.OcnError(f"no free port in {ports[0]}..{ports[-1]} on {addr} for the unique-id exchange")
        srv.listen(world)
        srv.settimeout(timeout)
        for _ in range(world - 1):
            conn, _peer = srv.accept()
            conn.sendall(_BOOT_MAGIC + payload)
            conn.close()
        srv.close()
        return payload
    deadline = time.time() + timeout
    want = len(_BOOT_MAGIC) + 128
    while time.time() < deadline:
        for port in ports:
            try:
                c = socket.create_connection((addr, port), timeout=2.0)
            except OSError:
                continue
            try:
                c.settimeout(5.0)
                data = b""
                while len(data) < want:
                    chunk = c.recv(want - len(data))
                    if not chunk:
                        break
                    data += chunk
            except OSError:
                data = b""
            finally:
                c.close()
            if len(data) == want and data.startswith(_BOOT_MAGIC):
                return data[len(_BOOT_MAGIC):]
        time.sleep(0.05)
    raise _lib.OcnError(f"rank {rank}: no unique id from rank 0 at {addr}:{ports[0]}..{ports[-1]}")


from .models import NonhydrostaticModel as _NonhydrostaticModel   # noqa: E402


class LibraryDistributedModel(_NonhydrostaticModel):
    """NonhydrostaticModel(grid::DistributedRectilinearGrid; ...) whose time-step -- halo exchanges, interior / buffer split,
    distributed pressure solve -- runs inside the library (ocn_dist_model_create). Same host API as NonhydrostaticModel:
    ocn.time_step, ocn.set_model (LOCAL interior arrays), ocn.update_state, model.clock, Simulation."""

    def __init__(self, grid, **kwargs):
        if not isinstance(grid.ctx, Distributed):
            raise TypeError("LibraryDistributedModel needs a grid on a `Distributed` architecture (communicator inside the library)")
        self.ctx = grid.ctx
        super().__init__(grid, **kwargs)

    def _create_handle(self, grid, ntracers):
        h = C.c_void_p()
        bounded = getattr(grid, "global_x_topology", Periodic) is Bounded
        Rx, Ry = getattr(grid, "partition", (grid.ctx.world, 1))
        if Ry > 1:
            sx = (C.c_int * Rx)(*grid.local_sizes)
            sy = (C.c_int * Ry)(*grid.local_sizes_y)
            _lib.check(_lib.lib().ocn_dist_model_create_pencil(C.byref(h), grid.local.handle, ntracers, grid.ctx.handle, float(grid.Lx_global),
                                                               float(grid.Ly_global), Rx, Ry, sx, sy, 1 if bounded else 0,
                                                               1 if getattr(grid, "global_y_topology", Periodic) is Bounded else 0))
        elif bounded:
            sizes = (C.c_int * len(grid.local_sizes))(*grid.local_sizes)
            _lib.check(_lib.lib().ocn_dist_model_create_partition(C.byref(h), grid.local.handle, ntracers, grid.ctx.handle,
                                                                  float(grid.Lx_global), sizes, 1))
        elif getattr(grid, "irregular", False):          # Periodic x, Nx % R != 0: the remainder on the last rank
            sizes = (C.c_int * len(grid.local_sizes))(*grid.local_sizes)
            _lib.check(_lib.lib().ocn_dist_model_create_sizes(C.byref(h), grid.local.handle, ntracers, grid.ctx.handle,
                                                              float(grid.Lx_global), sizes))
        else:
            _lib.check(_lib.lib().ocn_dist_model_create(C.byref(h), grid.local.handle, ntracers, grid.ctx.handle, float(grid.Lx_global)))
        return h

    def max_abs_divergence(self):
        v = C.c_double()
        _lib.check(_lib.lib().ocn_dist_model_max_abs_divergence(self.handle, C.byref(v)))
        return v.value
