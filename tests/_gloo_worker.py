"""Worker for tests/test_sharding_cpu.py: one rank of a world_size-2 gloo group.

Runs the product's sharding driver (shard_bounds + ShardedFIR.filter_local_host +
GlooTransport halo exchange) with the CPU oracle injected as the per-shard kernel
(the oracle is the CHECKER here: the assertion compares the sharded result with the
oracle applied to the whole vector)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sk_dsp_comm_amd import sharding
    from oracle import oracle as orc

    rng = np.random.default_rng(99)  # same stream on every rank
    n, P = 20011, 257
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).astype(np.complex64)
    b = rng.standard_normal(P) / 16

    tr = sharding.GlooTransport()
    assert (tr.rank, tr.world) == (rank, world)
    bounds = sharding.shard_bounds(n, world)
    s0, s1 = bounds[rank]
    fir = sharding.ShardedFIR(b, tr, dtype=np.complex64,
                              kernel=lambda xl, hist: orc.fir_filter(b, xl, hist=hist))
    y_local = fir.filter_local_host(x[s0:s1])
    y_full = orc.fir_filter(b, x)
    err = float(np.max(np.abs(y_local - y_full[s0:s1])) / np.max(np.abs(y_full)))
    # the halo really came from the neighbour (rank 0: zeros)
    hist = tr.halo_exchange_host(x[s0:s1], P - 1)
    ok_hist = bool(np.array_equal(hist, x[s0 - (P - 1):s0])) if rank > 0 else bool(np.all(hist == 0))
    tmax = tr.allreduce_max(float(rank + 1))
    tr.barrier()
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write("%r %r %r %d %d\n" % (err, ok_hist, tmax, s0, s1))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
