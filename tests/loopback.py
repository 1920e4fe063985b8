"""TEST-ONLY in-process 'communicator': R virtual ranks = R Python threads sharing ONE GPU, with the collectives of
DistributedContext implemented as stream-ordered tensor copies between the ranks' buffers. It exercises the product's
DeviceBackend (pack / unpack kernels, distributed FFT stages, interior/strip tendency split) on the 1-GPU box, where RCCL
cannot run two ranks on the same device. The real multi-GPU path differs only in who moves the bytes (RCCL)."""
import threading


class LoopbackWorld:
    def __init__(self, R, torch, arch):
        self.R, self.torch, self.arch = R, torch, arch
        self.barrier_obj = threading.Barrier(R)
        self.slots = [None] * R
        self.vals = [0.0] * R

    def context(self, rank):
        return LoopbackContext(self, rank)


class LoopbackContext:
    def __init__(self, world, rank):
        self.w = world
        self.rank, self.world = rank, world.R
        self.torch, self.arch = world.torch, world.arch
        self.device = world.torch.device("cuda", 0)
        self.west, self.east = (rank - 1) % world.R, (rank + 1) % world.R
        self.partitioned = world.R > 1

    def exchange_start(self, ws, es, wr, er):
        w = self.w
        w.slots[self.rank] = (ws, es)
        w.barrier_obj.wait()                 # every rank has SUBMITTED its pack kernel (same stream => ordered)
        er.copy_(w.slots[self.east][0])      # east neighbour's west slab -> my east halo
        wr.copy_(w.slots[self.west][1])      # west neighbour's east slab -> my west halo
        w.barrier_obj.wait()
        return []

    @staticmethod
    def exchange_wait(reqs):
        pass

    def exchange(self, ws, es, wr, er):
        self.exchange_start(ws, es, wr, er)

    def all_to_all(self, recv, send):
        w = self.w
        w.slots[self.rank] = send
        w.barrier_obj.wait()
        n = send.numel() // self.world
        for s in range(self.world):
            recv[s * n:(s + 1) * n].copy_(w.slots[s][self.rank * n:(self.rank + 1) * n])
        w.barrier_obj.wait()

    def all_gather(self, gathered, payload):
        w = self.w
        w.slots[self.rank] = payload
        w.barrier_obj.wait()
        n = payload.numel()
        for s in range(self.world):
            gathered[s * n:(s + 1) * n].copy_(w.slots[s])
        w.barrier_obj.wait()

    def allreduce_max(self, value):
        w = self.w
        w.vals[self.rank] = float(value)
        w.barrier_obj.wait()
        m = max(w.vals)
        w.barrier_obj.wait()
        return m

    def barrier(self):
        self.w.barrier_obj.wait()


class PointerLoopbackWorld:
    """The same rendezvous for the IN-LIBRARY partitioned model (ocn_dist_create_transport): the collectives arrive as raw device
    addresses from the library's own orchestration; copies are the library's stream-ordered device copies (all virtual ranks share
    the library's one compute stream, the barriers order the SUBMISSION of pack kernels and copies)."""

    def __init__(self, R, lib):
        self.R, self.lib = R, lib
        self.barrier_obj = threading.Barrier(R)
        self.slots = [None] * R
        self.vals = [0.0] * R

    def collectives(self, rank):
        return PointerLoopback(self, rank)


class PointerLoopback:
    def __init__(self, world, rank):
        self.w, self.rank, self.R = world, rank, world.R
        self.west, self.east = (rank - 1) % world.R, (rank + 1) % world.R

    def _copy(self, dst, src, ndoubles):
        rc = self.w.lib.ocn_memcpy_d2d(dst, src, 8 * ndoubles)
        assert rc == 0, rc

    def exchange_start(self, ws, es, wr, er, n):
        w = self.w
        w.slots[self.rank] = (ws, es)
        w.barrier_obj.wait()
        self._copy(er, w.slots[self.east][0], n)      # east neighbour's west slab -> my east halo
        self._copy(wr, w.slots[self.west][1], n)      # west neighbour's east slab -> my west halo
        w.barrier_obj.wait()

    def exchange_wait(self):
        pass

    def exchange_peers(self, peer_lo, peer_hi, lo_send, hi_send, lo_recv, hi_recv, n):
        """pencil partitions: what leaves through the low side lands in peer_lo's HIGH halo"""
        w = self.w
        w.slots[self.rank] = (lo_send, hi_send)
        w.barrier_obj.wait()
        self._copy(hi_recv, w.slots[peer_hi][0], n)
        self._copy(lo_recv, w.slots[peer_lo][1], n)
        w.barrier_obj.wait()

    def all_to_all(self, send, recv, n):
        w = self.w
        w.slots[self.rank] = send
        w.barrier_obj.wait()
        for s in range(self.R):
            self._copy(recv + 8 * n * s, w.slots[s] + 8 * n * self.rank, n)
        w.barrier_obj.wait()

    def all_gather(self, send, recv, n):
        w = self.w
        w.slots[self.rank] = send
        w.barrier_obj.wait()
        for s in range(self.R):
            self._copy(recv + 8 * n * s, w.slots[s], n)
        w.barrier_obj.wait()

    def all_to_all_group(self, peers, send, recv, n):
        """MPI.Alltoallv! on a sub-communicator (the pencil transposes): chunk q of `recv` is the chunk of peers[q]'s send buffer that is
        addressed to this rank, i.e. the one at this rank's position in the (common) group list. Every rank of the world is inside
        such a call at the same time, each in its own group, so the world barrier still orders submissions."""
        w = self.w
        w.slots[self.rank] = send
        w.barrier_obj.wait()
        me = peers.index(self.rank)
        for q, peer in enumerate(peers):
            self._copy(recv + 8 * n * q, w.slots[peer] + 8 * n * me, n)
        w.barrier_obj.wait()

    def allreduce_max(self, value):
        w = self.w
        w.vals[self.rank] = float(value)
        w.barrier_obj.wait()
        m = max(w.vals)
        w.barrier_obj.wait()
        return m
