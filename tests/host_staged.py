"""TEST-ONLY transport for the in-library partitioned model (ocn_dist_create_transport): REAL separate processes that share ONE card,
every collective staged through host memory and run over gloo (RCCL refuses two ranks on one device). Same ranks, buffers and call
order as the RCCL transport; timings mean nothing."""
import numpy as np


class HostStagedCollectives:
    def __init__(self, torch, dist, lib, rank, world):
        self.torch, self.dist, self.lib, self.rank, self.world = torch, dist, lib, rank, world
        self.west, self.east = (rank - 1) % world, (rank + 1) % world

    def _host(self, ptr, n):
        a = np.empty(n, dtype=np.float64)
        assert self.lib.ocn_memcpy_d2h(a.ctypes.data, ptr, 8 * n) == 0      # synchronises the compute stream first
        return self.torch.from_numpy(a)

    def _device(self, ptr, t):
        a = np.ascontiguousarray(t.numpy())
        assert self.lib.ocn_memcpy_h2d(ptr, a.ctypes.data, 8 * a.size) == 0

    def exchange_start(self, ws, es, wr, er, n):
        d = self.dist
        hws, hes = self._host(ws, n), self._host(es, n)
        hwr, her = self.torch.empty_like(hws), self.torch.empty_like(hes)
        ops = [d.P2POp(d.isend, hws, self.west, tag=1), d.P2POp(d.irecv, her, self.east, tag=1),
               d.P2POp(d.isend, hes, self.east, tag=2), d.P2POp(d.irecv, hwr, self.west, tag=2)]
        for req in d.batch_isend_irecv(ops):
            req.wait()
        self._device(wr, hwr)
        self._device(er, her)

    def exchange_wait(self):
        pass

    def all_to_all(self, send, recv, n):
        d, R = self.dist, self.world
        h = self._host(send, n * R).reshape(R, n)
        out = self.torch.empty_like(h)
        reqs, bufs = [], {}
        for r in range(R):
            if r == self.rank:
                out[r].copy_(h[r])
            else:
                reqs.append(d.isend(h[r].contiguous(), r, tag=10 + self.rank))
                bufs[r] = self.torch.empty(n, dtype=self.torch.float64)
                reqs.append(d.irecv(bufs[r], r, tag=10 + r))
        for req in reqs:
            req.wait()
        for r, b in bufs.items():
            out[r].copy_(b)
        self._device(recv, out.reshape(-1))

    def all_gather(self, send, recv, n):
        h = self._host(send, n)
        out = self.torch.empty(self.world * n, dtype=self.torch.float64)
        self.dist.all_gather_into_tensor(out, h)
        self._device(recv, out)

    def allreduce_max(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())
