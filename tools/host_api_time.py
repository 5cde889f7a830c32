"""Where the time of the NumPy-in / NumPy-out API goes (PCIe-inclusive; never bench.py's `value`).
    python tools/host_api_time.py        (on a GPU box, from the repo root)"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, 'scikit-dsp-comm_amd')
sys.path.insert(0, '.')
import bench  # noqa: E402
from sk_dsp_comm_amd import _ffi, multirate_helper as mrh, config  # noqa: E402

b = bench.firwin_lowpass(1024, 0.2)
f = mrh.multirate_FIR(b)
rng = np.random.default_rng(0)
n = 1 << 24
x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).astype(np.complex64)
f.filter(x[:100000])


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


for strict in (False, True, False):
    config.strict_dtype = strict
    ms, y = best(lambda: f.filter(x))
    print("multirate_FIR.filter, 2^24 c64, strict_dtype=%s: %.1f ms -> %.0f MS/s (result %s)" % (strict, ms, n / ms / 1e3, y.dtype))

# long vector: single staged copy vs the chunk pipeline (H2D of chunk k+1 | kernels of chunk k | D2H of chunk k-1)
n2 = 1 << 26
x2 = np.tile(x, 4)
for strict in (False, True):
    config.strict_dtype = strict
    for pipe, lg in ((0, 22), (1, 24), (1, 23), (1, 22), (1, 21)):
        _ffi.set_option("host_pipeline", pipe); _ffi.set_option("host_chunk_log2", lg)
        ms, y = best(lambda: f.filter(x2), reps=3)
        print("multirate_FIR.filter, 2^26 c64 (512 MiB), strict_dtype=%s, pipeline=%d chunk 2^%d: %.1f ms -> %.0f MS/s" % (strict, pipe, lg, ms, n2 / ms / 1e3))
_ffi.set_option("host_pipeline", 1); _ffi.set_option("host_chunk_log2", 22)
config.strict_dtype = False
del x2, y

# the pieces, through the C ABI directly
k = _ffi.FirKernel(b, _ffi.C64)
L = _ffi.load()
y = np.empty_like(x)
ms, _ = best(lambda: _ffi.check(L.skdsp_fir_filter(ctypes.c_void_p(k.h), _ffi._ptr(x), n, _ffi._ptr(y))))
print("skdsp_fir_filter (host pointers, output array already touched): %.1f ms" % ms)
ms, _ = best(lambda: np.empty_like(x).fill(0))
print("first touch of a fresh 128 MiB output array (np.empty + fill): %.1f ms" % ms)
xd = _ffi.DeviceArray(n, np.complex64)
yd = _ffi.DeviceArray(n, np.complex64)
ms, _ = best(lambda: xd.write(x))
print("H2D 128 MiB: %.1f ms" % ms)
ms, _ = best(lambda: (k.filter_dev(xd, yd), _ffi.sync()))
print("kernel: %.2f ms" % ms)
ms, _ = best(lambda: _ffi.check(L.skdsp_memcpy_d2h(_ffi._ptr(y), ctypes.c_void_p(yd.ptr), y.nbytes)))
print("D2H 128 MiB into a touched array: %.1f ms" % ms)
ms, _ = best(lambda: x.astype(np.complex128))
print("astype(complex128) of the result (strict_dtype): %.1f ms" % ms)
