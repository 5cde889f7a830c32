"""ctypes binding of libskdsp_hip.so (the C ABI declared in include/skdsp.h).

No PyTorch, no CPU fallback: if the library or a HIP device is missing the calls
raise -- the product path must fail loudly rather than silently run elsewhere.
"""
import ctypes
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libskdsp_hip.so"

F32, C64, F64, C128 = 0, 1, 2, 3
FIR_AUTO, FIR_DIRECT, FIR_OLS = 0, 1, 2

_NP_OF = {F32: np.float32, C64: np.complex64, F64: np.float64, C128: np.complex128}
_CODE_OF = {np.dtype(np.float32): F32, np.dtype(np.complex64): C64, np.dtype(np.float64): F64,
            np.dtype(np.complex128): C128}

# every symbol include/skdsp.h declares (tests check the built library exports all of them)
SYMBOLS = [
    "skdsp_init", "skdsp_shutdown", "skdsp_debug_path", "skdsp_device_count", "skdsp_device_info", "skdsp_last_error", "skdsp_version",
    "skdsp_set_option", "skdsp_get_option", "skdsp_init_devices", "skdsp_slot_count", "skdsp_host_chunk_plan",
    "skdsp_host_alloc", "skdsp_host_free", "skdsp_malloc", "skdsp_free", "skdsp_memcpy_h2d", "skdsp_memcpy_d2h", "skdsp_memcpy_d2d", "skdsp_memset",
    "skdsp_sync", "skdsp_timer_start", "skdsp_timer_stop", "skdsp_last_kernel_ms", "skdsp_fill_noise_dev",
    "skdsp_fir_create", "skdsp_fir_set_algo", "skdsp_fir_get_algo", "skdsp_fir_filter", "skdsp_fir_filter_dev",
    "skdsp_fir_filter_rows", "skdsp_fir_filter_rows_dev", "skdsp_fir_filter_sharded",
    "skdsp_fir_up", "skdsp_fir_up_dev", "skdsp_fir_dn", "skdsp_fir_dn_dev", "skdsp_fir_updn", "skdsp_fir_updn_dev",
    "skdsp_sos_create", "skdsp_tf_create", "skdsp_tf2sos", "skdsp_iir_filter", "skdsp_iir_filter_dev", "skdsp_iir_up",
    "skdsp_iir_up_dev", "skdsp_iir_dn", "skdsp_iir_dn_dev", "skdsp_iir_state_len", "skdsp_iir_filter_state_dev",
    "skdsp_iir_filter_rows", "skdsp_iir_filter_rows_dev", "skdsp_sos_par_info", "skdsp_iir_sequential",
    "skdsp_sos_filter", "skdsp_sos_up", "skdsp_sos_dn",
    "skdsp_upsample", "skdsp_upsample_dev", "skdsp_downsample", "skdsp_downsample_dev", "skdsp_set_wide_output", "skdsp_destroy",
    "skdsp_dist_unique_id", "skdsp_dist_init", "skdsp_dist_shutdown", "skdsp_dist_comm_count", "skdsp_dist_barrier",
    "skdsp_dist_allreduce_max", "skdsp_dist_allreduce_sum", "skdsp_dist_sendrecv", "skdsp_dist_allgather", "skdsp_dist_halo_exchange", "skdsp_fir_filter_shard_dev",
]

_lib = None
_lock = threading.Lock()


class SkdspError(RuntimeError):
    pass


def lib_path():
    # SKDSP_LIB: developer override to A/B an alternative build of the same library
    return os.environ.get("SKDSP_LIB") or os.path.join(_HERE, LIB_NAME)


def load():
    """dlopen the library and declare prototypes.  Does NOT touch the GPU."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise SkdspError(
                "%s not found next to the package: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C scikit-dsp-comm_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
        L = ctypes.CDLL(path)
        vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        pvp = ctypes.POINTER(ctypes.c_void_p)
        L.skdsp_last_error.restype = ctypes.c_char_p
        L.skdsp_version.restype = ctypes.c_char_p
        L.skdsp_init.argtypes = [ci]
        if hasattr(L, "skdsp_debug_path"):   # (absent from older builds of the library that SKDSP_LIB may point at for A/B timing)
            L.skdsp_debug_path.argtypes = [ctypes.c_char_p, ci, ci]
        L.skdsp_init_devices.argtypes = [ctypes.POINTER(ci), ci]
        p64 = ctypes.POINTER(i64)
        L.skdsp_host_chunk_plan.argtypes = [i64, ci, ci, i64, ci, i64, p64, p64, p64, p64, p64, p64]
        L.skdsp_set_option.argtypes = [ctypes.c_char_p, ci]
        L.skdsp_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ci)]
        L.skdsp_device_info.argtypes = [ctypes.c_char_p, ci, ctypes.POINTER(ci), ctypes.POINTER(i64), ctypes.POINTER(ci)]
        L.skdsp_host_alloc.argtypes = [pvp, i64]
        L.skdsp_host_free.argtypes = [vp]
        L.skdsp_malloc.argtypes = [pvp, i64]
        L.skdsp_free.argtypes = [vp]
        for f in (L.skdsp_memcpy_h2d, L.skdsp_memcpy_d2h, L.skdsp_memcpy_d2d):
            f.argtypes = [vp, vp, i64]
        L.skdsp_memset.argtypes = [vp, ci, i64]
        L.skdsp_timer_stop.argtypes = [ctypes.POINTER(ctypes.c_float)]
        L.skdsp_fill_noise_dev.argtypes = [vp, i64, ci, ctypes.c_uint64, i64]
        L.skdsp_fir_create.argtypes = [vp, ci, ci, ci, pvp]
        L.skdsp_fir_set_algo.argtypes = [vp, ci]
        L.skdsp_fir_get_algo.argtypes = [vp, i64, ctypes.POINTER(ci)]
        L.skdsp_fir_filter.argtypes = [vp, vp, i64, vp]
        L.skdsp_fir_filter_sharded.argtypes = [vp, vp, i64, vp, ci]
        L.skdsp_fir_filter_dev.argtypes = [vp, vp, i64, i64, vp]
        L.skdsp_fir_filter_rows.argtypes = [vp, vp, i64, i64, vp]
        L.skdsp_fir_filter_rows_dev.argtypes = [vp, vp, i64, i64, i64, i64, vp]
        L.skdsp_fir_up.argtypes = [vp, vp, i64, ci, vp]
        L.skdsp_fir_up_dev.argtypes = [vp, vp, i64, i64, ci, vp]
        L.skdsp_fir_dn.argtypes = [vp, vp, i64, ci, vp]
        L.skdsp_fir_dn_dev.argtypes = [vp, vp, i64, i64, ci, vp]
        L.skdsp_fir_updn.argtypes = [vp, vp, i64, ci, ci, vp]
        L.skdsp_fir_updn_dev.argtypes = [vp, vp, i64, i64, ci, ci, vp]
        L.skdsp_sos_create.argtypes = [vp, ci, ci, pvp]
        L.skdsp_tf_create.argtypes = [vp, ci, vp, ci, ci, pvp]
        L.skdsp_tf2sos.argtypes = [vp, ci, vp, ci, vp, ctypes.POINTER(ci)]
        L.skdsp_iir_filter.argtypes = [vp, vp, i64, vp]
        L.skdsp_iir_filter_dev.argtypes = [vp, vp, i64, vp]
        L.skdsp_iir_state_len.argtypes = [vp, ctypes.POINTER(ci)]
        L.skdsp_iir_filter_state_dev.argtypes = [vp, vp, i64, vp, vp, vp]
        L.skdsp_iir_filter_rows.argtypes = [vp, vp, i64, i64, vp]
        L.skdsp_iir_filter_rows_dev.argtypes = [vp, vp, i64, i64, i64, i64, vp]
        L.skdsp_sos_par_info.argtypes = [vp, ci, vp, ctypes.POINTER(ci)]
        if hasattr(L, "skdsp_iir_sequential"):
            L.skdsp_iir_sequential.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_double)]
        L.skdsp_iir_up.argtypes = [vp, vp, i64, ci, vp]
        L.skdsp_iir_up_dev.argtypes = [vp, vp, i64, ci, vp]
        L.skdsp_iir_dn.argtypes = [vp, vp, i64, ci, vp]
        L.skdsp_iir_dn_dev.argtypes = [vp, vp, i64, ci, vp]
        L.skdsp_sos_filter.argtypes = [vp, vp, i64, vp]
        L.skdsp_sos_up.argtypes = [vp, vp, i64, ci, vp]
        L.skdsp_sos_dn.argtypes = [vp, vp, i64, ci, vp]
        L.skdsp_last_kernel_ms.restype = ctypes.c_double
        L.skdsp_upsample.argtypes = [vp, i64, ci, ci, vp]
        L.skdsp_upsample_dev.argtypes = [vp, i64, ci, ci, ctypes.c_double, vp]
        L.skdsp_downsample.argtypes = [vp, i64, ci, ci, ci, vp]
        L.skdsp_downsample_dev.argtypes = [vp, i64, ci, ci, ci, vp]
        L.skdsp_destroy.argtypes = [vp]
        L.skdsp_dist_unique_id.argtypes = [vp]
        L.skdsp_dist_init.argtypes = [ci, ci, vp]
        L.skdsp_dist_comm_count.argtypes = [ctypes.POINTER(ci)]
        L.skdsp_dist_allreduce_max.argtypes = [ctypes.POINTER(ctypes.c_double)]
        L.skdsp_dist_allreduce_sum.argtypes = [ctypes.POINTER(ctypes.c_double)]
        L.skdsp_set_wide_output.argtypes = [vp, ci]
        L.skdsp_dist_allgather.argtypes = [vp, vp, i64]
        L.skdsp_dist_sendrecv.argtypes = [vp, ci, vp, ci, i64]
        L.skdsp_dist_allgather.argtypes = [vp, vp, i64]
        L.skdsp_dist_halo_exchange.argtypes = [vp, i64, i64, ci]
        L.skdsp_fir_filter_shard_dev.argtypes = [vp, vp, i64, vp]
        _lib = L
        return _lib


_EXC = {-1: ValueError, -2: SkdspError, -3: MemoryError, -4: SkdspError, -5: SkdspError, -6: NotImplementedError}


def check(rc):
    if rc != 0:
        msg = load().skdsp_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, SkdspError)("skdsp[%d]: %s" % (rc, msg))


def init(device=None):
    """Bind this process: one GPU (default: SKDSP_DEVICE, else LOCAL_RANK, else 0), or -- when SKDSP_DEVICES is set to
    "all" or a list like "0,1,2,3" and no launcher gave this process a rank -- one slot per listed GPU (init_devices)."""
    L = load()
    if device is None:
        devs = os.environ.get("SKDSP_DEVICES")
        if devs and "LOCAL_RANK" not in os.environ and "SKDSP_DEVICE" not in os.environ:
            return init_devices(devs)
        device = int(os.environ.get("SKDSP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    check(L.skdsp_init(int(device)))


def init_devices(devices="all"):
    """One slot per GPU: the host-array FIR calls (multirate_FIR.filter / .up / .dn on long NumPy vectors) then use all
    of them from one process.  devices: "all", "0,1,2" or a sequence of indices; returns the number of slots."""
    L = load()
    if isinstance(devices, str):
        devices = list(range(L.skdsp_device_count())) if devices.strip().lower() == "all" else [int(v) for v in devices.split(",") if v.strip()]
    devices = [int(d) for d in devices]
    if not devices:
        raise SkdspError("no HIP device available: the MI355X path has no CPU fallback")
    arr = (ctypes.c_int * len(devices))(*devices)
    check(L.skdsp_init_devices(arr, len(devices)))
    return L.skdsp_slot_count()


def host_chunk_plan(n, L=1, M=1, hist=0, chunk_log2=24):
    """[(in_begin, in_end, in_hist, out_begin, out_end)] of every chunk the host-pointer entry points would use."""
    lib = load()
    v = [ctypes.c_int64(0) for _ in range(6)]
    check(lib.skdsp_host_chunk_plan(int(n), int(L), int(M), int(hist), int(chunk_log2), 0, *[ctypes.byref(a) for a in v]))
    out = []
    for k in range(v[0].value):
        check(lib.skdsp_host_chunk_plan(int(n), int(L), int(M), int(hist), int(chunk_log2), k, *[ctypes.byref(a) for a in v]))
        out.append(tuple(a.value for a in v[1:]))
    return out


def set_option(name, value):
    """Run-time switch of the library (struct Options in csrc/skdsp_internal.hpp); returns the previous value."""
    L = load()
    old = ctypes.c_int(0)
    check(L.skdsp_get_option(name.encode(), ctypes.byref(old)))
    check(L.skdsp_set_option(name.encode(), int(value)))
    return old.value


def get_option(name):
    v = ctypes.c_int(0)
    check(load().skdsp_get_option(name.encode(), ctypes.byref(v)))
    return v.value


class option:
    """with _ffi.option("iir_planar", 1): ...  -- temporary switch (tests, A/B timing)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def debug_path(clear=True):
    """Names of the kernel families this thread's calls launched since the last clear (tests: which engine did a shape reach?)."""
    buf = ctypes.create_string_buffer(256)
    check(load().skdsp_debug_path(buf, 256, 1 if clear else 0))
    return buf.value.decode().split(",") if buf.value else []


def device_info():
    L = load()
    name = ctypes.create_string_buffer(256)
    cus, hbm, clk = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int(0)
    check(L.skdsp_device_info(name, 256, ctypes.byref(cus), ctypes.byref(hbm), ctypes.byref(clk)))
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value, "clock_khz": clk.value}


def code_of(dtype):
    try:
        return _CODE_OF[np.dtype(dtype)]
    except KeyError:
        raise TypeError("unsupported signal dtype %r (float32/complex64/float64/complex128)" % (dtype,))


def np_of(code):
    return _NP_OF[code]


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a.size else ctypes.c_void_p(0)


# --------------------------------------------------------------------------
# device-resident arrays (used by bench.py / sharding; not part of the reference surface)
# --------------------------------------------------------------------------
class DeviceArray:
    """n samples of `dtype` in HBM with `headroom` samples of valid-able space in front
    (the halo / history region: x[-headroom..-1])."""

    def __init__(self, n, dtype, headroom=0):
        L = load()
        self.n = int(n)
        self.code = code_of(dtype)
        self.dtype = np.dtype(dtype)
        esz = self.dtype.itemsize
        # keep x[0] 256-byte aligned whatever the headroom is
        self._pad = ((int(headroom) * esz + 255) // 256) * 256
        self.headroom = int(headroom)
        base = ctypes.c_void_p(0)
        check(L.skdsp_malloc(ctypes.byref(base), self._pad + self.n * esz + 256))
        self._base = base.value
        self.ptr = self._base + self._pad
        self._fin = weakref.finalize(self, _free_dev, self._base)
        if self._pad:
            check(L.skdsp_memset(ctypes.c_void_p(self._base), 0, self._pad))

    @classmethod
    def from_host(cls, a, headroom=0):
        a = np.ascontiguousarray(a)
        d = cls(a.size, a.dtype, headroom)
        check(load().skdsp_memcpy_h2d(ctypes.c_void_p(d.ptr), _ptr(a), a.nbytes))
        return d

    def write(self, a, at=0):
        """Copy host samples to x[at .. at+len(a)); negative `at` addresses the headroom."""
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if at < -self.headroom or at + a.size > self.n:
            raise ValueError("write outside the device array")
        if a.size:
            check(load().skdsp_memcpy_h2d(ctypes.c_void_p(self.ptr + at * self.dtype.itemsize), _ptr(a), a.nbytes))

    def to_host(self, start=0, count=None):
        start = int(start)
        count = self.n - start if count is None else int(count)
        out = np.empty(count, dtype=self.dtype)
        esz = self.dtype.itemsize
        check(load().skdsp_memcpy_d2h(_ptr(out), ctypes.c_void_p(self.ptr + start * esz), out.nbytes))
        return out

    def window(self, start, count):
        """A non-owning view of samples [start, start + count) (kernels that write a slice of a larger device buffer)."""
        if start < 0 or start + count > self.n:
            raise ValueError("window outside the device array")
        v = object.__new__(DeviceArray)
        v.n, v.code, v.dtype, v.headroom, v._pad = int(count), self.code, self.dtype, 0, 0
        v._base = None
        v.ptr = self.ptr + int(start) * self.dtype.itemsize
        v._fin = lambda: None
        v._owner = self   # keeps the allocation alive
        return v

    def fill_noise(self, seed, first_index=0):
        check(load().skdsp_fill_noise_dev(ctypes.c_void_p(self.ptr), self.n, self.code, int(seed), int(first_index)))
        return self

    def free(self):
        self._fin()


def _free_dev(base):
    try:
        load().skdsp_free(ctypes.c_void_p(base))
    except Exception:
        pass


def sync():
    check(load().skdsp_sync())


def timer_start():
    check(load().skdsp_timer_start())


def timer_stop():
    ms = ctypes.c_float(0.0)
    check(load().skdsp_timer_stop(ctypes.byref(ms)))
    return float(ms.value)


# --------------------------------------------------------------------------
# handles
# --------------------------------------------------------------------------
def _destroy(h):
    try:
        load().skdsp_destroy(ctypes.c_void_p(h))
    except Exception:
        pass


_WIDE = {np.dtype(np.float32): np.dtype(np.float64), np.dtype(np.complex64): np.dtype(np.complex128)}


class _PinnedBlock:
    """One page-locked block; exposes its bytes through __array_interface__ and goes back to the pool when the last
    ndarray view of it dies."""

    def __init__(self, pool, ptr, nbytes):
        self.pool, self.ptr, self.nbytes = pool, ptr, nbytes
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            self.pool._give_back(self.ptr, self.nbytes)
        except Exception:
            pass


class PinnedPool:
    """Result arrays of long host-array calls come from recycled page-locked blocks.

    A copy back into a fresh np.empty array pays for the population of its pages and their first device access
    (31 ms per 512 MiB instead of 12, and another 26 ms when NumPy unmaps it again: tools/host_pipe_time.py); a
    page-locked block receives the DMA at the link rate and is handed to the next call of the same size when its
    array is garbage collected.  The arrays are ordinary writable ndarrays whose .base chain ends in the block: they do
    NOT own their memory (ndarray.resize refuses them; np.copy gives an owning array).
    max_bytes bounds what the pool keeps for reuse; max_live_bytes bounds the page-locked memory in the hands of live
    result arrays plus the pool (beyond it results are ordinary np.empty arrays); 0 switches the pool off.
    The lock is re-entrant and nothing is allocated while it is held: a block's finaliser (_give_back) can run inside
    any allocation -- including one made by this class -- when the cyclic collector fires."""

    GRANULE = 2 << 20

    def __init__(self, max_bytes=4 << 30, min_bytes=32 << 20, max_live_bytes=None):
        self.max_bytes, self.min_bytes = int(max_bytes), int(min_bytes)
        self.max_live_bytes = int(max_live_bytes) if max_live_bytes is not None else 4 * self.max_bytes
        self._free = {}      # rounded size -> [ptr]
        self._kept = 0
        self._live = 0       # bytes of blocks currently wrapped by result arrays
        self._lock = threading.RLock()

    def empty(self, count, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(count) * dtype.itemsize
        if self.max_bytes <= 0 or nbytes < self.min_bytes:
            return np.empty(count, dtype=dtype)
        size = -(-nbytes // self.GRANULE) * self.GRANULE
        ptr = None
        with self._lock:
            lst = self._free.get(size)
            if lst:
                ptr = lst.pop()
                self._kept -= size
                self._live += size
            elif self._live + self._kept + size > self.max_live_bytes:
                size = 0   # cap on page-locked host memory reached
            else:
                self._live += size   # reserved before the allocation below (which runs outside the lock)
        if size == 0:
            return np.empty(count, dtype=dtype)
        if ptr is None:
            p = ctypes.c_void_p(0)
            try:
                check(load().skdsp_host_alloc(ctypes.byref(p), size))
            except Exception:
                with self._lock:
                    self._live -= size
                return np.empty(count, dtype=dtype)   # no page-locked memory left: an ordinary array
            ptr = p.value
        block = _PinnedBlock(self, ptr, size)
        return np.asarray(block)[:nbytes].view(dtype)

    def _give_back(self, ptr, size):
        keep = False
        with self._lock:
            self._live -= size
            lst = self._free.get(size)
            if self._kept + size <= self.max_bytes and lst is not None:
                lst.append(ptr)          # (list.append of an int allocates nothing the collector tracks)
                self._kept += size
                keep = True
        if keep:
            return
        if self._kept + size <= self.max_bytes:
            fresh = [ptr]                # first block of this size: the list is built OUTSIDE the lock
            with self._lock:
                if self._kept + size <= self.max_bytes:
                    cur = self._free.get(size)
                    if cur is None:
                        self._free[size] = fresh
                    else:
                        cur.append(ptr)
                    self._kept += size
                    return
        load().skdsp_host_free(ctypes.c_void_p(ptr))

    def trim(self):
        """Release every block the pool holds for reuse."""
        with self._lock:
            old, self._free, self._kept = self._free, {}, 0
        for lst in old.values():
            for p in lst:
                load().skdsp_host_free(ctypes.c_void_p(p))


result_pool = PinnedPool(int(os.environ.get("SKDSP_PINNED_POOL_BYTES", str(4 << 30))))


class _HostCalls:
    """Shared by FirKernel / IirKernel: output allocation for the host-pointer entry points.
    wide=True asks the library for float64/complex128 results from a float32/complex64 handle
    (widened on the device: skdsp_set_wide_output)."""
    _wide_state = False
    _call_lock = None

    def _host(self, count, dtype, wide, call):
        """One host-pointer call: the per-handle result-width switch and the call it applies to form one critical section,
        so an object shared between threads cannot get the other thread's width."""
        if self._call_lock is None:
            self.__dict__.setdefault("_call_lock", threading.Lock())
        with self._call_lock:
            y = self._out(count, dtype, wide)
            call(y)
            return y

    def _out(self, count, dtype, wide):
        dtype = np.dtype(dtype)
        wide = bool(wide) and dtype in _WIDE
        if wide != self._wide_state:
            check(load().skdsp_set_wide_output(ctypes.c_void_p(self.h), int(wide)))
            self._wide_state = wide
        return result_pool.empty(count, _WIDE[dtype] if wide else dtype)


class FirKernel(_HostCalls):
    """A device FIR handle for one (taps, signal dtype) pair."""

    def __init__(self, taps, code):
        L = load()
        taps = np.asarray(taps)
        self.taps_complex = bool(np.iscomplexobj(taps))
        t = np.ascontiguousarray(taps, dtype=np.complex128 if self.taps_complex else np.float64)
        self.ntaps = int(t.size)
        self.code = code
        h = ctypes.c_void_p(0)
        check(L.skdsp_fir_create(_ptr(t), self.ntaps, int(self.taps_complex), code, ctypes.byref(h)))
        self.h = h.value
        self._fin = weakref.finalize(self, _destroy, self.h)

    def set_algo(self, algo):
        check(load().skdsp_fir_set_algo(ctypes.c_void_p(self.h), int(algo)))

    def algo_for(self, n):
        a = ctypes.c_int(0)
        check(load().skdsp_fir_get_algo(ctypes.c_void_p(self.h), int(n), ctypes.byref(a)))
        return a.value

    # host vectors ------------------------------------------------------
    def filter(self, x, wide=False):
        return self._host(x.size, x.dtype, wide, lambda y: check(load().skdsp_fir_filter(ctypes.c_void_p(self.h), _ptr(x), x.size, _ptr(y))))

    def filter_sharded(self, x, ngpu=0, wide=False):
        """.filter(x) over the first ngpu slots bound by init_devices (0: all): contiguous sample blocks, halos from the host array."""
        return self._host(x.size, x.dtype, wide,
                          lambda y: check(load().skdsp_fir_filter_sharded(ctypes.c_void_p(self.h), _ptr(x), x.size, _ptr(y), int(ngpu))))

    def up(self, x, L, wide=False):
        return self._host(x.size * L, x.dtype, wide, lambda y: check(load().skdsp_fir_up(ctypes.c_void_p(self.h), _ptr(x), x.size, int(L), _ptr(y))))

    def dn(self, x, M, wide=False):
        return self._host(x.size // M, x.dtype, wide, lambda y: check(load().skdsp_fir_dn(ctypes.c_void_p(self.h), _ptr(x), x.size, int(M), _ptr(y))))

    def updn(self, x, L, M, wide=False):
        return self._host((x.size * L) // M, x.dtype, wide, lambda y: check(load().skdsp_fir_updn(ctypes.c_void_p(self.h), _ptr(x), x.size, int(L), int(M), _ptr(y))))

    # device vectors ----------------------------------------------------
    def filter_dev(self, xd, yd, n=None, n_hist=0):
        n = xd.n if n is None else n
        check(load().skdsp_fir_filter_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, n_hist, ctypes.c_void_p(yd.ptr)))

    def up_dev(self, xd, yd, L, n=None, n_hist=0):
        n = xd.n if n is None else n
        check(load().skdsp_fir_up_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, n_hist, int(L), ctypes.c_void_p(yd.ptr)))

    def dn_dev(self, xd, yd, M, n=None, n_hist=0):
        n = xd.n if n is None else n
        check(load().skdsp_fir_dn_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, n_hist, int(M), ctypes.c_void_p(yd.ptr)))

    def updn_dev(self, xd, yd, L, M, n=None, n_hist=0):
        n = xd.n if n is None else n
        check(load().skdsp_fir_updn_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, n_hist, int(L), int(M),
                                         ctypes.c_void_p(yd.ptr)))

    def filter_rows(self, x2, wide=False):
        """x2: C-contiguous (rows, n): every row filtered from rest in ONE call (one pitched copy each way, one launch)."""
        rows, n = x2.shape
        y = self._host(x2.size, x2.dtype, wide, lambda y: check(load().skdsp_fir_filter_rows(ctypes.c_void_p(self.h), _ptr(x2), n, rows, _ptr(y))))
        return y.reshape(rows, n)

    def filter_rows_dev(self, xd, yd, n, rows, x_stride=None, y_stride=None):
        check(load().skdsp_fir_filter_rows_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, rows, n if x_stride is None else x_stride,
                                                n if y_stride is None else y_stride, ctypes.c_void_p(yd.ptr)))

    def filter_shard_dev(self, xd, yd, n=None):
        n = xd.n if n is None else n
        check(load().skdsp_fir_filter_shard_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, ctypes.c_void_p(yd.ptr)))


class IirKernel(_HostCalls):
    """A device IIR handle: cascaded biquads (sos) or a transfer function (b, a)."""

    def __init__(self, code, sos=None, b=None, a=None):
        L = load()
        self.code = code
        h = ctypes.c_void_p(0)
        if sos is not None:
            s = np.ascontiguousarray(sos, dtype=np.float64)
            check(L.skdsp_sos_create(_ptr(s), int(s.shape[0]), code, ctypes.byref(h)))
        else:
            bb = np.ascontiguousarray(np.atleast_1d(b), dtype=np.float64)
            aa = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
            check(L.skdsp_tf_create(_ptr(bb), bb.size, _ptr(aa), aa.size, code, ctypes.byref(h)))
        self.h = h.value
        self._fin = weakref.finalize(self, _destroy, self.h)
        seq, spread = ctypes.c_int(0), ctypes.c_double(0.0)
        if hasattr(L, "skdsp_iir_sequential"):
            check(L.skdsp_iir_sequential(ctypes.c_void_p(self.h), ctypes.byref(seq), ctypes.byref(spread)))
        self.sequential, self.spread = bool(seq.value), spread.value
        if self.sequential:
            import logging
            logging.getLogger("sk_dsp_comm_amd").warning(
                "IIR cascade of %d sections is ill-conditioned: two float64 evaluations of it (sections in another order) differ by %.1e of "
                "the output; it runs the reference's own sample-by-sample recursion on the GPU (exact, but at a host core's speed)",
                int(np.asarray(sos).shape[0]) if sos is not None else -1, self.spread)

    def filter(self, x, wide=False):
        return self._host(x.size, x.dtype, wide, lambda y: check(load().skdsp_iir_filter(ctypes.c_void_p(self.h), _ptr(x), x.size, _ptr(y))))

    def filter_rows(self, x2, wide=False):
        """x2: C-contiguous (rows, n): every row filtered from rest in ONE call (one copy each way, one launch where the
        parallel-form scan applies)."""
        rows, n = x2.shape
        y = self._host(x2.size, x2.dtype, wide, lambda y: check(load().skdsp_iir_filter_rows(ctypes.c_void_p(self.h), _ptr(x2), n, rows, _ptr(y))))
        return y.reshape(rows, n)

    def filter_rows_dev(self, xd, yd, n, rows, x_stride=None, y_stride=None):
        check(load().skdsp_iir_filter_rows_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, rows, n if x_stride is None else x_stride,
                                                n if y_stride is None else y_stride, ctypes.c_void_p(yd.ptr)))

    def up(self, x, L, wide=False):
        return self._host(x.size * L, x.dtype, wide, lambda y: check(load().skdsp_iir_up(ctypes.c_void_p(self.h), _ptr(x), x.size, int(L), _ptr(y))))

    def dn(self, x, M, wide=False):
        return self._host(x.size // M, x.dtype, wide, lambda y: check(load().skdsp_iir_dn(ctypes.c_void_p(self.h), _ptr(x), x.size, int(M), _ptr(y))))

    def filter_dev(self, xd, yd, n=None):
        n = xd.n if n is None else n
        check(load().skdsp_iir_filter_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, ctypes.c_void_p(yd.ptr)))

    def up_dev(self, xd, yd, L, n=None):
        n = xd.n if n is None else n
        check(load().skdsp_iir_up_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, int(L), ctypes.c_void_p(yd.ptr)))

    def dn_dev(self, xd, yd, M, n=None):
        n = xd.n if n is None else n
        check(load().skdsp_iir_dn_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, int(M), ctypes.c_void_p(yd.ptr)))

    def state_len(self):
        k = ctypes.c_int(0)
        check(load().skdsp_iir_state_len(ctypes.c_void_p(self.h), ctypes.byref(k)))
        return k.value

    def filter_state_dev(self, xd, yd, n=None, zi=None, want_zf=True):
        """Device-resident block streaming: y_dev[0:n] and (returned) the state after sample n-1.
        zi: float64 vector of state_len() entries, None = rest."""
        n = xd.n if n is None else n
        zf = np.zeros(self.state_len()) if want_zf else None
        zi_p = None
        if zi is not None:
            zi = np.ascontiguousarray(zi, dtype=np.float64)
            if zi.size != self.state_len():
                raise ValueError("zi must have %d entries" % self.state_len())
            zi_p = _ptr(zi)
        check(load().skdsp_iir_filter_state_dev(ctypes.c_void_p(self.h), ctypes.c_void_p(xd.ptr), n, zi_p,
                                                 _ptr(zf) if want_zf else None, ctypes.c_void_p(yd.ptr)))
        return zf

    def filter_state(self, x, zi=None):
        """Block streaming from host memory: (y, zf)."""
        x = np.ascontiguousarray(x)
        xd = DeviceArray.from_host(x)
        yd = DeviceArray(x.size, x.dtype)
        try:
            zf = self.filter_state_dev(xd, yd, x.size, zi)
            y = yd.to_host()
        finally:
            xd.free()
            yd.free()
        return y, zf


def sos_par_info(sos):
    """Host only: the partial-fraction expansion the parallel-form scan runs for this cascade (c0, sections (a1, a2, r0, r1),
    cancellation factor, impulse-response error against the cascade, accepted)."""
    s = np.ascontiguousarray(sos, dtype=np.float64).reshape(-1, 6)
    ns = s.shape[0]
    out = np.zeros(5 + 4 * ns)
    ok = ctypes.c_int(0)
    check(load().skdsp_sos_par_info(_ptr(s), ns, _ptr(out), ctypes.byref(ok)))
    return {"c0": out[0], "sections": out[1:1 + 4 * ns].reshape(ns, 4).copy(), "kappa": out[1 + 4 * ns], "ir_err": out[2 + 4 * ns],
            "accepted": bool(ok.value),
            # the float32 from-rest states (7 - 8 biquads, float32 / complex64 signals): worst probe error on 128- / 96-sample chunks, admitted below 5e-7
            "v32_err": out[3 + 4 * ns], "v32_err_t96": out[4 + 4 * ns], "v32_admitted": bool(ok.value) and out[3 + 4 * ns] <= 5e-7}


def tf2sos(b, a):
    """The (b, a) -> second-order-sections factorisation skdsp_tf_create applies (host only)."""
    bb = np.ascontiguousarray(np.atleast_1d(b), dtype=np.float64)
    aa = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
    sos = np.zeros((12, 6))
    ns = ctypes.c_int(0)
    check(load().skdsp_tf2sos(_ptr(bb), bb.size, _ptr(aa), aa.size, _ptr(sos), ctypes.byref(ns)))
    return sos[:ns.value].copy()


def upsample(x, L):
    y = np.empty(x.size * L, dtype=x.dtype)
    check(load().skdsp_upsample(_ptr(x), x.size, int(L), code_of(x.dtype), _ptr(y)))
    return y


def downsample(x, M, p):
    y = np.empty(x.size // M, dtype=x.dtype)
    check(load().skdsp_downsample(_ptr(x), x.size, int(M), int(p), code_of(x.dtype), _ptr(y)))
    return y
