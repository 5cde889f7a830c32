"""sigsys.downsample on the device, 2^26 samples, a few dtypes and strides -- one line per run; alternate with SKDSP_LIB=<another build> for a same-box A/B
(profiles/r06/experiments/ab_downsample.txt): python tools/ab_downsample.py"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from sk_dsp_comm_amd import _ffi
_ffi.init(0)
n = 1 << 26
L = _ffi.load()
out = []
for dt, M in ((np.complex64, 3), (np.complex64, 2), (np.float32, 3), (np.complex128, 3), (np.complex64, 8)):
    xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n // M, dt)
    for rep in range(3):
        _ffi.timer_start()
        for _ in range(60): _ffi.check(L.skdsp_downsample_dev(ctypes.c_void_p(xd.ptr), n, M, 0, _ffi.code_of(dt), ctypes.c_void_p(yd.ptr)))
        ms = _ffi.timer_stop() / 60
    out.append("%s/%d %.4f" % (np.dtype(dt).name, M, ms))
    xd.free(); yd.free()
print(os.environ.get("SKDSP_LIB", "in-tree"), " | ".join(out), flush=True)
