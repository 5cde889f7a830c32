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
sys.path.insert(0, os.path.join(ROOT, "tests"))


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

    from _gloo_transport import GlooTransport
    tr = GlooTransport()
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
    # ---- sharded IIR: exact state hand-off (scipy's sosfilt as the per-shard kernel) ----
    from scipy import signal
    def sos_kernel(sos):
        def run(xl, zi):
            zi = np.zeros((sos.shape[0], 2)) if zi is None else zi
            return signal.sosfilt(sos, xl, zi=zi)
        return run
    iir_err = 0.0
    xr = rng.standard_normal(n)
    for sos in (signal.ellip(6, 0.5, 60, [0.2, 0.4], btype="bandpass", output="sos"),   # decays: short head
                np.array([[1.0, 0.0, 0.0, 1.0, -1.0, 0.0]]),                           # integrator: full re-filter
                signal.butter(4, 0.002, output="sos")):                                  # slow decay
        for xs in (xr, xr + 1j * xr[::-1]):
            iir = sharding.ShardedIIR(sos, tr, dtype=xs.dtype, kernel=sos_kernel(sos), head_quantum=512)
            yl = iir.filter_local_host(xs[s0:s1])
            yf = signal.sosfilt(sos, xs)
            iir_err = max(iir_err, float(np.max(np.abs(yl - yf[s0:s1])) / np.max(np.abs(yf))))
    tmax = tr.allreduce_max(float(rank + 1))
    tr.barrier()
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write("%r %r %r %d %d %r\n" % (err, ok_hist, tmax, s0, s1, iir_err))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
