"""Host-buffer transport over torch.distributed (gloo) for the world_size-2 CPU tests of
sk_dsp_comm_amd.sharding (test infrastructure: the product transport is RcclTransport)."""
import numpy as np


class GlooTransport:
    """Host-buffer transport over torch.distributed (gloo).  The process group must
    already be initialised by the caller (tests / a TCP launcher)."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def barrier(self):
        self._dist.barrier()

    def allreduce_max(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def allgather_state(self, vec):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64).copy())
        out = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t)
        return np.stack([o.numpy() for o in out])

    def halo_exchange_host(self, x_local, n_halo):
        """Send my last n_halo samples right, receive the left neighbour's; rank 0 gets zeros."""
        import torch
        dist = self._dist
        x_local = np.ascontiguousarray(x_local)
        if n_halo > x_local.size:
            raise ValueError("halo of %d samples needs a shard of at least that many (got %d)" % (n_halo, x_local.size))
        hist = np.zeros(n_halo, dtype=x_local.dtype)
        if n_halo == 0 or self.world == 1:
            return hist
        reqs = []
        if self.rank + 1 < self.world:
            tail = torch.from_numpy(np.ascontiguousarray(x_local[x_local.size - n_halo:]).view(np.uint8).copy())
            reqs.append(dist.isend(tail, self.rank + 1))
        if self.rank > 0:
            buf = torch.empty(n_halo * x_local.dtype.itemsize, dtype=torch.uint8)
            dist.recv(buf, self.rank - 1)
            hist = buf.numpy().view(x_local.dtype).copy()
        for r in reqs:
            r.wait()
        return hist
