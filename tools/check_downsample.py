"""Developer check + timing of sigsys.downsample on the device (resample.hip): exactness over dtypes / strides / phases / ragged ends, then 2^26 samples.
Run on the GPU box: python tools/check_downsample.py"""
import sys, time, ctypes
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from sk_dsp_comm_amd import _ffi, sigsys as ss
_ffi.init(0)
rng = np.random.default_rng(0)
# exactness incl. ragged ends and phases
for dt in (np.float32, np.complex64, np.float64, np.complex128):
    for M in (2, 3, 4, 5, 7, 8, 16):
        for n in (4096 * M + 5, 100003, 65536 * 3 + 1):
            for p in (0, M - 1):
                x = rng.standard_normal(n).astype(dt) if np.dtype(dt).kind != "c" else (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dt)
                y = ss.downsample(x, M, p)
                ref = x[p::M][: n // M] if False else x[0:(n // M) * M].reshape(-1, M)[:, p]
                assert y.shape == ref.shape and np.array_equal(y, ref), (dt, M, n, p)
print("exact ok")
n = 1 << 26
for dt in (np.complex64, np.float32, np.complex128):
    for M in (2, 3, 4, 8):
        xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n // M, dt)
        L = _ffi.load()
        for rep in range(2):
            _ffi.timer_start()
            for _ in range(40): _ffi.check(L.skdsp_downsample_dev(ctypes.c_void_p(xd.ptr), n, M, 0, _ffi.code_of(dt), ctypes.c_void_p(yd.ptr)))
            ms = _ffi.timer_stop() / 40
        isz = np.dtype(dt).itemsize
        print("%-10s M=%d: %.4f ms  %.2f TB/s algorithmic (n + n/M)" % (np.dtype(dt).name, M, ms, isz * (n + n // M) / ms / 1e9))
        xd.free(); yd.free()
