"""C-level timing of the host-pointer FIR entry point on a long vector (output array already touched / fresh):
python tools/host_pipe_time.py"""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, 'scikit-dsp-comm_amd'); sys.path.insert(0, '.')
import bench
from sk_dsp_comm_amd import _ffi
b = bench.firwin_lowpass(1024, 0.2)
k = _ffi.FirKernel(b, _ffi.C64)
L = _ffi.load()
n = 1 << 26
rng = np.random.default_rng(0)
x = np.tile(((rng.standard_normal(1 << 22) + 1j * rng.standard_normal(1 << 22)) / np.sqrt(2)).astype(np.complex64), 16)
y = np.empty_like(x); y.fill(0)
def call(yy): _ffi.check(L.skdsp_fir_filter(ctypes.c_void_p(k.h), _ffi._ptr(x), n, _ffi._ptr(yy)))
def best(fn, reps=4):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, ts
for pipe, lg in ((0, 24), (1, 24), (1, 23), (1, 22), (1, 25)):
    _ffi.set_option("host_pipeline", pipe); _ffi.set_option("host_chunk_log2", lg)
    ms, ts = best(lambda: call(y))
    yp = _ffi.result_pool.empty(n, np.complex64)
    ms3, ts3 = best(lambda: call(yp))
    def fresh():
        yy = np.empty_like(x); call(yy)
    ms2, _ = best(fresh)
    print("pipeline=%d chunk 2^%d: touched pageable y %.1f ms   page-locked y %.1f ms (%s)   fresh y (+ free) %.1f ms" % (pipe, lg, ms, ms3, " ".join("%.1f" % (t * 1e3) for t in ts3), ms2), flush=True)
    del yp
