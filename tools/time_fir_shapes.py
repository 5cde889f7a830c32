"""Time overlap-save / direct FIR shapes on one GPU (2^26 samples): python tools/time_fir_shapes.py
   complex64 1024 taps (headline), float32 1024 / 300 / 127 taps, complex64 .dn(3) 512 taps, float32 .dn(4) 1024 taps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
n = 1 << 26
_ffi.init(0)
def timeit(step, steps=300):
    for _ in range(150): step()
    _ffi.sync(); _ffi.timer_start()
    for _ in range(steps): step()
    return _ffi.timer_stop() / steps
tag = os.path.basename(os.environ.get("SKDSP_LIB", "default"))
for dt, ntaps, M in ((np.complex64, 1024, 1), (np.float32, 1024, 1), (np.float32, 300, 1), (np.float32, 127, 1), (np.complex64, 512, 3), (np.float32, 1024, 4)):
    b = bench.firwin_lowpass(ntaps, 0.2)
    k = _ffi.FirKernel(b, _ffi.code_of(dt))
    xd = _ffi.DeviceArray(n, dt, headroom=ntaps).fill_noise(1)
    yd = _ffi.DeviceArray(n // M, dt)
    ms = timeit((lambda: k.filter_dev(xd, yd)) if M == 1 else (lambda: k.dn_dev(xd, yd, M)))
    esz = np.dtype(dt).itemsize
    bytes_ = esz * n + esz * (n // M)
    print("%s %-9s %4d taps M=%d: %.4f ms  %.2f TB/s  %.1f %% of 8 TB/s" % (tag, np.dtype(dt).name, ntaps, M, ms, bytes_ / ms / 1e9, bytes_ / ms / 1e6 / 80), flush=True)
    xd.free(); yd.free()
