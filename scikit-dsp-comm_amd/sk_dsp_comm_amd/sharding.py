"""Sample-block sharding of a long FIR across the GPUs of one node.

One process per GPU.  Rank r owns the contiguous samples [start_r, stop_r) of the
signal; the only exchange the FIR needs is the Ntaps-1 input samples preceding a
shard, which rank r-1 sends from the END of its own shard (rank 0 uses zeros == the
zero initial state of lfilter).  Outputs need no exchange: rank r's outputs are the
samples [start_r, stop_r) of the full-length result.

The SOS IIR shards the same way (ShardedIIR): every rank filters its block from rest, the
end states are all-gathered (2*n_sections doubles per rank) and folded locally into each rank's
true initial state (s_{k+1} = f_k + A^{n_k} s_k), and each rank re-filters only the head of its block -- as far as its true initial state
is still visible -- from that state.

Transports
  RcclTransport  device buffers, RCCL send/recv over xGMI inside libskdsp_hip.so
                 (the production path; rendezvous of the RCCL unique id through a
                 file because the product does not depend on torch)
  (tests/_gloo_transport.py holds a host-buffer transport over torch.distributed/gloo with the
   same methods: it drives these classes in the world_size-2 CPU tests; it is not product code.)
"""
import os
import time

import numpy as np

from . import _ffi


def shard_bounds(n, world, multiple=1):
    """Contiguous shards [(start, stop)] covering [0, n); every boundary is a multiple of
    `multiple` (use M for a decimator so that output phase 0 stays aligned)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    units = -(-n // multiple)
    base, rem = divmod(units, world)
    bounds, pos = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        start = min(pos * multiple, n)
        pos += cnt
        bounds.append((start, min(pos * multiple, n)))
    return bounds


def env_rank_world():
    """(rank, world, local_rank) from the launcher's environment (torchrun-compatible)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


class FileRendezvous:
    """Tiny single-node rendezvous: rank 0 publishes a blob, everyone reads it.

    The directory name is unique per launch: it is keyed by the launcher's pid and
    start time (all ranks share one parent) plus MASTER_PORT, so stale files from an
    earlier run cannot be picked up."""

    def __init__(self, rank, world, tag=None, root=None, timeout=300.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        if tag is None:
            ppid = os.getppid()
            try:
                with open("/proc/%d/stat" % ppid) as f:
                    start = f.read().rsplit(")", 1)[1].split()[19]
            except Exception:
                start = "0"
            tag = "%s_%d_%s" % (os.environ.get("MASTER_PORT", "0"), ppid, start)
        root = root or os.environ.get("SKDSP_RDZV_DIR", "/tmp")
        self.dir = os.path.join(root, "skdsp_rdzv_" + tag)
        os.makedirs(self.dir, exist_ok=True)

    def _wait(self, path):
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > self.timeout:
                raise TimeoutError("rendezvous: %s did not appear within %.0f s" % (path, self.timeout))
            time.sleep(0.01)

    def broadcast(self, name, blob=None):
        path = os.path.join(self.dir, name)
        if self.rank == 0:
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "wb") as f:
                f.write(blob)
            os.rename(tmp, path)  # atomic publish
            return blob
        self._wait(path)
        with open(path, "rb") as f:
            return f.read()

    def barrier(self, name):
        open(os.path.join(self.dir, "%s.%d" % (name, self.rank)), "wb").close()
        for r in range(self.world):
            self._wait(os.path.join(self.dir, "%s.%d" % (name, r)))

    def cleanup(self):
        """Every rank signs off; rank 0 waits for all sign-offs, then removes the directory
        (the other ranks do not wait, so nothing can be deleted under a reader)."""
        open(os.path.join(self.dir, "bye.%d" % self.rank), "wb").close()
        if self.rank != 0:
            return
        for r in range(self.world):
            self._wait(os.path.join(self.dir, "bye.%d" % r))
        for fn in os.listdir(self.dir):
            try:
                os.remove(os.path.join(self.dir, fn))
            except OSError:
                pass
        try:
            os.rmdir(self.dir)
        except OSError:
            pass


class RcclTransport:
    """RCCL communicator owned by libskdsp_hip.so; one rank per GPU."""

    def __init__(self, rank=None, world=None, local_rank=None, rdzv=None):
        er, ew, el = env_rank_world()
        self.rank = er if rank is None else rank
        self.world = ew if world is None else world
        local_rank = el if local_rank is None else local_rank
        if "SKDSP_DEVICE" in os.environ:  # explicit override (e.g. several ranks on one GPU for debugging)
            local_rank = int(os.environ["SKDSP_DEVICE"])
        _ffi.init(local_rank)
        L = _ffi.load()
        import ctypes
        if self.world > 1:
            rdzv = rdzv or FileRendezvous(self.rank, self.world)
            blob = None
            if self.rank == 0:
                buf = ctypes.create_string_buffer(128)
                _ffi.check(L.skdsp_dist_unique_id(buf))
                blob = buf.raw
            blob = rdzv.broadcast("rccl_id", blob)
            _ffi.check(L.skdsp_dist_init(self.rank, self.world, ctypes.c_char_p(blob)))
            rdzv.barrier("rccl_up")
            self._rdzv = rdzv
        else:
            _ffi.check(L.skdsp_dist_init(0, 1, None))
            self._rdzv = None

    def comm_count(self):
        """Ranks in the live RCCL communicator (ncclCommCount); 0 when a 1-rank job built none."""
        import ctypes
        k = ctypes.c_int(0)
        _ffi.check(_ffi.load().skdsp_dist_comm_count(ctypes.byref(k)))
        return k.value

    def barrier(self):
        _ffi.check(_ffi.load().skdsp_dist_barrier())

    def allreduce_max(self, v):
        import ctypes
        d = ctypes.c_double(float(v))
        _ffi.check(_ffi.load().skdsp_dist_allreduce_max(ctypes.byref(d)))
        return d.value

    def allreduce_sum(self, v):
        import ctypes
        d = ctypes.c_double(float(v))
        _ffi.check(_ffi.load().skdsp_dist_allreduce_sum(ctypes.byref(d)))
        return d.value

    def halo_exchange_dev(self, xd, n, n_halo):
        import ctypes
        _ffi.check(_ffi.load().skdsp_dist_halo_exchange(ctypes.c_void_p(xd.ptr), n, n_halo, xd.code))

    def allgather_state(self, vec):
        """Every rank contributes a small float64 vector; returns the (world, len) table.  One RCCL
        all-gather on the compute stream through persistent staging buffers."""
        import ctypes
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        st = getattr(self, "_stage", None)
        if st is None or st[0].n < vec.size:
            cap = max(64, vec.size)
            st = self._stage = (_ffi.DeviceArray(cap, np.float64), _ffi.DeviceArray(cap * self.world, np.float64))
        st[0].write(vec)
        _ffi.check(_ffi.load().skdsp_dist_allgather(ctypes.c_void_p(st[0].ptr), ctypes.c_void_p(st[1].ptr), vec.nbytes))
        return st[1].to_host(0, vec.size * self.world).reshape(self.world, vec.size)

    def close(self):
        _ffi.load().skdsp_dist_shutdown()
        if self._rdzv is not None:
            self._rdzv.cleanup()


def hip_fir_kernel(fir_kernel):
    """kernel(x_local, hist) -> y_local on the GPU for host-buffer transports."""
    def run(x_local, hist):
        x_local = np.ascontiguousarray(x_local)
        xd = _ffi.DeviceArray(x_local.size, x_local.dtype, headroom=max(len(hist), 1))
        esz = x_local.dtype.itemsize
        import ctypes
        L = _ffi.load()
        _ffi.check(L.skdsp_memcpy_h2d(ctypes.c_void_p(xd.ptr), ctypes.c_void_p(x_local.ctypes.data), x_local.nbytes))
        if len(hist):
            h = np.ascontiguousarray(hist, dtype=x_local.dtype)
            _ffi.check(L.skdsp_memcpy_h2d(ctypes.c_void_p(xd.ptr - len(h) * esz), ctypes.c_void_p(h.ctypes.data), h.nbytes))
        yd = _ffi.DeviceArray(x_local.size, x_local.dtype)
        fir_kernel.filter_dev(xd, yd, n_hist=len(hist))
        y = yd.to_host()
        xd.free()
        yd.free()
        return y
    return run


class ShardedFIR:
    """multirate_FIR.filter on a sample-block shard (multirate_helper.py:104-109 applied
    to samples [start_r, stop_r) of a longer vector).

    filter_local_dev : device-resident shard + RcclTransport (production / bench.py)
    filter_local_host: host shard + any host transport and kernel (tests, TCP clusters)
    """

    def __init__(self, b, transport, dtype=np.complex64, kernel=None):
        self.b = np.asarray(b)
        self.ntaps = len(self.b)
        self.halo = self.ntaps - 1
        self.transport = transport
        self.dtype = np.dtype(dtype)
        self._fir = None
        self._kernel = kernel

    def _hip(self):
        if self._fir is None:
            self._fir = _ffi.FirKernel(self.b, _ffi.code_of(self.dtype))
        return self._fir

    def new_shard_buffer(self, n_local):
        """Device buffer for a shard with headroom for the halo in front of x[0]."""
        return _ffi.DeviceArray(n_local, self.dtype, headroom=max(self.halo, 1))

    # ---- device-resident shards (production path) ---------------------------------------
    # Every method: (1) the transport fills x[-halo..-1] from the left neighbour (zeros on rank
    # 0 == the zero initial state of lfilter), (2) the single-GPU kernel runs with n_hist=halo.
    # Outputs need no exchange: rank r produces the outputs that belong to its input block.
    def _check(self, n_local, halo):
        if self.transport.world > 1 and n_local < halo:
            raise ValueError("shard of %d samples is shorter than the %d-sample halo" % (n_local, halo))

    def filter_local_dev(self, xd, yd, n_local=None):
        """.filter: halo Ntaps-1; outputs [start_r, stop_r)."""
        n_local = xd.n if n_local is None else n_local
        self._check(n_local, self.halo)
        if isinstance(self.transport, RcclTransport):
            # one library call: the halo crosses xGMI on a second stream while the overlap-save tiles
            # that do not need it already run (skdsp_fir_filter_shard_dev)
            self._hip().filter_shard_dev(xd, yd, n_local)
            return
        self.transport.halo_exchange_dev(xd, n_local, self.halo)
        self._hip().filter_dev(xd, yd, n_local, n_hist=self.halo)

    def up_halo(self, L):
        """input samples of history .up(x, L) needs: ceil((Ntaps-1)/L)"""
        return -(-(self.ntaps - 1) // L)

    def up_local_dev(self, xd, yd, L, n_local=None):
        """.up(x, L): outputs [start_r*L, stop_r*L)."""
        n_local = xd.n if n_local is None else n_local
        halo = self.up_halo(L)
        self._check(n_local, halo)
        self.transport.halo_exchange_dev(xd, n_local, halo)
        self._hip().up_dev(xd, yd, L, n_local, n_hist=halo)

    def dn_local_dev(self, xd, yd, M, n_local=None):
        """.dn(x, M): shard starts must be multiples of M (shard_bounds(n, world, multiple=M));
        outputs [start_r/M, stop_r/M)."""
        n_local = xd.n if n_local is None else n_local
        self._check(n_local, self.halo)
        self.transport.halo_exchange_dev(xd, n_local, self.halo)
        self._hip().dn_dev(xd, yd, M, n_local, n_hist=self.halo)

    def updn_local_dev(self, xd, yd, L, M, n_local=None):
        """downsample(.up(x, L), M): shard starts must be multiples of M/gcd(L, M);
        outputs [start_r*L/M, stop_r*L/M)."""
        n_local = xd.n if n_local is None else n_local
        halo = self.up_halo(L)
        self._check(n_local, halo)
        self.transport.halo_exchange_dev(xd, n_local, halo)
        self._hip().updn_dev(xd, yd, L, M, n_local, n_hist=halo)

    def filter_local_host(self, x_local):
        x_local = np.ascontiguousarray(x_local, dtype=self.dtype)
        hist = self.transport.halo_exchange_host(x_local, self.halo)
        kernel = self._kernel or hip_fir_kernel(self._hip())
        return kernel(x_local, hist)


def sos_state_matrix(sos):
    """State-transition matrix A (zero input) of the DF2T biquad cascade in scipy.signal.sosfilt's
    zi coordinates (n_sections x 2, flattened): state_after = A @ state_before for x = 0."""
    sos = np.atleast_2d(np.asarray(sos, dtype=np.float64))
    ns = sos.shape[0]
    D = 2 * ns
    A = np.zeros((D, D))
    for j in range(D):
        z = np.zeros(D)
        z[j] = 1.0
        x = 0.0
        for k in range(ns):
            b0, b1, b2, _, a1, a2 = sos[k]
            y = b0 * x + z[2 * k]
            z[2 * k] = b1 * x - a1 * y + z[2 * k + 1]
            z[2 * k + 1] = b2 * x - a2 * y
            x = y
        A[:, j] = z
    return A


def hip_iir_kernel(iir_kernel):
    """kernel(x_local, zi) -> (y_local, zf) on the GPU for host-buffer transports; zi / zf are
    (n_sections, 2) arrays (complex for a complex signal), None = rest."""
    def run(x_local, zi):
        cplx = np.iscomplexobj(x_local)
        flat = None
        if zi is not None:
            zi = np.asarray(zi)
            flat = np.concatenate([zi.real.ravel(), zi.imag.ravel()]) if cplx else zi.real.ravel()
        y, out = iir_kernel.filter_state(x_local, flat)
        h = out.size // 2
        zf = (out[:h] + 1j * out[h:]).reshape(-1, 2) if cplx else out.reshape(-1, 2)
        return y, zf
    return run


class ShardedIIR:
    """multirate_IIR.filter (multirate_helper.py:169-174) on a sample-block shard: EXACT state
    hand-off, no approximation of the recursion.

      1. every rank filters its block from rest -> y0, end state f_r        (parallel, one pass)
      2. ONE all-gather of (n_r, f_r) (1 + 2*n_sections doubles per rank); each rank folds its
         own true initial state: s_0 = zi (rest), s_{k+1} = f_k + A^{n_k} s_k for k < r
      3. by linearity only the zero-input response of s_r is missing from y0; it is below
         `decay_tol` after K samples (max |A^K| <= decay_tol), so the first K samples are
         re-filtered from s_r; K = n_r (a second full pass) for a filter that never decays
         (integrators).
    """

    def __init__(self, sos, transport, dtype=np.float32, kernel=None, decay_tol=1e-18, head_quantum=4096):
        self.sos = np.atleast_2d(np.asarray(sos, dtype=np.float64))
        if self.sos.ndim != 2 or self.sos.shape[1] != 6:
            raise ValueError('sos array must be shape (n_sections, 6)')
        self.nsec = self.sos.shape[0]
        self.transport = transport
        self.dtype = np.dtype(dtype)
        self.cplx = self.dtype.kind == "c"
        self.decay_tol = float(decay_tol)
        self.head_quantum = int(head_quantum)
        self.A = sos_state_matrix(self.sos)
        self._pow = {}
        self._head = {}
        self._kernel = kernel
        self._iir = None

    def _hip(self):
        if self._iir is None:
            if self.nsec > 12:
                raise ValueError("ShardedIIR: at most 12 sections per device cascade")
            self._iir = _ffi.IirKernel(_ffi.code_of(self.dtype), sos=self.sos)
        return self._iir

    def _power(self, n):
        P = self._pow.get(n)
        if P is None:
            P = self._pow[n] = np.linalg.matrix_power(self.A, int(n))
        return P

    def head_length(self, n_local):
        """Smallest K (head_quantum * 2^k, capped at n_local) with max|A^K| <= decay_tol."""
        K = self._head.get(n_local)
        if K is None:
            K = self.head_quantum
            P = np.linalg.matrix_power(self.A, K)
            while K < n_local and np.max(np.abs(P)) > self.decay_tol:
                P = P @ P
                K *= 2
            K = self._head[n_local] = min(K, n_local)
        return K

    # ---- step 2 -------------------------------------------------------------------------
    def _pack(self, f, n_local):
        f = np.asarray(f).ravel()
        return np.concatenate([[float(n_local)], f.real, f.imag if self.cplx else []])

    def initial_state(self, table, rank, zi=None):
        """table: (world, 1 + D[*2]) rows [n_k, f_k] from the all-gather -> state before my block."""
        D = 2 * self.nsec
        ctype = np.complex128 if self.cplx else np.float64
        s = np.zeros(D, dtype=ctype) if zi is None else np.asarray(zi, dtype=ctype).ravel().copy()
        for k in range(rank):
            row = table[k]
            f = row[1:1 + D] + 1j * row[1 + D:1 + 2 * D] if self.cplx else row[1:1 + D]
            s = f + self._power(int(row[0])) @ s
        return s.reshape(self.nsec, 2)

    def exchange(self, f, n_local, zi=None):
        tr = self.transport
        table = tr.allgather_state(self._pack(f, n_local))
        return self.initial_state(table, tr.rank, zi)

    # ---- drivers ------------------------------------------------------------------------
    def filter_local_host(self, x_local, zi=None):
        """y for my block [start_r, stop_r); zi = state before sample 0 of the WHOLE signal."""
        x_local = np.ascontiguousarray(x_local, dtype=self.dtype)
        kernel = self._kernel or hip_iir_kernel(self._hip())
        y, f = kernel(x_local, None)
        s = self.exchange(f, x_local.size, zi)
        if np.any(s != 0) and x_local.size:
            K = self.head_length(x_local.size)
            y = np.array(y, copy=True)
            y[:K], _ = kernel(x_local[:K], s)
        return y

    def filter_local_dev(self, xd, yd, n_local=None, zi=None):
        """Device-resident shard: two launches of the scan (whole block from rest, head from s_r)."""
        n_local = xd.n if n_local is None else n_local
        k = self._hip()
        out = k.filter_state_dev(xd, yd, n_local, None)
        D = 2 * self.nsec
        f = out[:D] + 1j * out[D:] if self.cplx else out
        s = self.exchange(f, n_local, zi)
        if np.any(s != 0) and n_local:
            K = self.head_length(n_local)
            flat = np.concatenate([s.real.ravel(), s.imag.ravel()]) if self.cplx else s.ravel()
            k.filter_state_dev(xd, yd, K, flat, want_zf=False)
