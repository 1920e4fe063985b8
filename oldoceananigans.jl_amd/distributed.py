"""Distributed x-slab NonhydrostaticModel (reference: src/DistributedComputations/, src/Models/interleave_communication_and_computation.jl).

MI355X-first design: ONE process per GPU, everything behind the C ABI. The library owns the RCCL communicator and runs the partitioned
time-step itself: `Distributed` (ocn_dist_create / ocn_dist_create_transport), `DistributedRectilinearGrid` (the rank's slab of the global
grid: local_size, partition_coordinate, insert_connected_topology), `LibraryDistributedModel` (ocn_dist_model_create* ->
ocn_model_time_step). No torch in the process.

  * halo exchange (fill_halo_regions! of a partitioned field, halo_communication.jl:87-110,170-187): local y / z fills, then ONE
    send / recv pair with the two ring neighbours carrying all fields of the call (Hx x Py x Pz x nfields doubles per side -- corners
    ride along, communication_buffers.jl:53,71-76);
  * distributed pressure solve (distributed_fft_based_poisson_solver.jl:141-188): substructured x solve with one small all-gather
    (z Periodic) or the reference's two all-to-all transposes (z Bounded).

The host-orchestrated form of the same step over torch.distributed (the round-1 implementation, kept as a cross-check and as the
world_size-2 gloo CPU test's harness) lives in tests/host_orchestration.py -- it is not part of the product.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .grids import (Bounded, Center, Face, Flat, FullyConnected, LeftConnected, Periodic, RectilinearGrid, RightConnected,
                    _regular_coordinate)


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
        # `size` and `extent` list the non-Flat directions only, like RectilinearGrid's (Grids/input_validation.jl); full 3-tuples too
        flat = [t is Flat for t in topology]

        def expand(values, fill, what):
            values = tuple(values) if isinstance(values, (tuple, list)) else (values,)
            if len(values) == 3:
                return tuple(fill if f else v for v, f in zip(values, flat))
            if len(values) != 3 - sum(flat):
                raise ValueError(f"{what} must have {3 - sum(flat)} (non-Flat directions) or 3 entries")
            it = iter(values)
            return tuple(fill if f else next(it) for f in flat)
        size = expand(size, 1, "size")
        if extent is not None:
            ext = expand(extent, 1.0, "extent")
            x, y, z = (0.0, float(ext[0])), (0.0, float(ext[1])), (-float(ext[2]), 0.0)
        x, y, z = ((0.0, 1.0) if f else c for c, f in zip((x, y, z), flat))
        halo = expand(halo, 0, "halo")
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
        self.Ly_global = 1.0 if flat[1] else _regular_coordinate(y, self.global_size[1], "y")[1]
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
class _LocalView:
    def __init__(self, model):
        self.grid = model.grid.local
        self._model = model

    def fields(self):
        return self._model.fields()


def local_initial_state(model, fn):
    """evaluate a global initial-state generator on this rank's slab: `fn(ocn, model_like)` is called with an object whose
    grid.nodes() are the LOCAL node coordinates"""
    import sys
    return fn(sys.modules[__name__.rsplit(".", 1)[0]], _LocalView(model))




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
        if hasattr(collectives, "all_to_all_group"):    # pencil transposes: all-to-all inside a group given by its members' ranks
            cbs["all_to_all_group"] = T.ALL_TO_ALL_GROUP(guard(lambda u, peers, n, s_, r_, cnt, st: collectives.all_to_all_group([peers[q] for q in range(n)], s_, r_, cnt)))
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
