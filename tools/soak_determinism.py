"""Repeat-launch determinism soak of the FIR / resample engines: every engine's output of launch k must equal launch 0's bit for bit (the kernels have no
data-dependent summation order; a difference would be a race -- e.g. a barrier that was removed on an argument that does not hold).
Run on the GPU box: python tools/soak_determinism.py [repeats=200] [log2n=24]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = 1 << lg
_ffi.init(0)
xc = _ffi.DeviceArray(n, np.complex64).fill_noise(3)
xr = _ffi.DeviceArray(n, np.float32).fill_noise(4)
xz = _ffi.DeviceArray(n // 2, np.complex128).fill_noise(5)
yc = _ffi.DeviceArray(2 * n, np.complex64)
yr = _ffi.DeviceArray(2 * n, np.float32)
yz = _ffi.DeviceArray(n // 2, np.complex128)
K = lambda P, c, dt: _ffi.FirKernel(bench.firwin_lowpass(P, c), dt)
k1024c, k1024r, k1024z = K(1024, 0.2, _ffi.C64), K(1024, 0.2, _ffi.F32), K(1024, 0.2, _ffi.C128)
k4c, k4r = K(1024, 0.2, _ffi.C64), K(1024, 0.2, _ffi.F32)
k127, k512c, k43 = K(127, 0.2, _ffi.F32), K(512, 0.9 / 12, _ffi.C64), K(512, 0.225, _ffi.C64)
import ctypes
L = _ffi.load()
only = os.environ.get('SOAK_ONLY')
cases = [
    ("filter 1024 taps complex64 (overlap-save, two barriers per tile)", lambda: k1024c.filter_dev(xc, yc, n), yc, n),
    ("filter 1024 taps float32 (two real tiles)", lambda: k1024r.filter_dev(xr, yr, n), yr, n),
    ("filter 1024 taps complex128", lambda: k1024z.filter_dev(xz, yz, n // 2), yz, n // 2),
    ("dn by 4, 1024 taps complex64 (folded inverse)", lambda: k4c.dn_dev(xc, yc, 4, n), yc, n // 4),
    ("dn by 2, 1024 taps float32 (folded inverse, two real tiles)", lambda: k4r.dn_dev(xr, yr, 2, n), yr, n // 2),
    ("dn by 3, 1024 taps complex64", lambda: k4c.dn_dev(xc, yc, 3, n), yc, n // 3),
    ("up by 4, 1024 taps complex64 (replicated spectrum)", lambda: k4c.up_dev(xc, yc, 4, n // 4), yc, n),
    ("up by 2, 1024 taps float32", lambda: k4r.up_dev(xr, yr, 2, n // 2), yr, n),
    ("filter 127 taps float32 (matrix pipe, tiles stored as runs)", lambda: k127.filter_dev(xr, yr, n), yr, n),
    ("up by 12, 512 taps complex64 (matrix pipe, wave pairs)", lambda: k512c.up_dev(xc, yc, 12, n // 12), yc, (n // 12) * 12),
    ("dn by 12, 512 taps complex64 (matrix pipe, lag split)", lambda: k512c.dn_dev(xc, yc, 12, n), yc, n // 12),
    ("up 4 / dn 3, 512 taps complex64 (matrix pipe)", lambda: k43.updn_dev(xc, yc, 4, 3, n // 4), yc, ((n // 4) * 4) // 3),
    ("downsample by 3 complex64", lambda: _ffi.check(L.skdsp_downsample_dev(ctypes.c_void_p(xc.ptr), n, 3, 1, _ffi.C64, ctypes.c_void_p(yc.ptr))), yc, n // 3),
]
bad = 0
for name, fn, yd, cnt in cases:
    if only and only not in name:
        continue
    _ffi.debug_path()   # (cleared)
    fn(); _ffi.sync()
    first = zlib.crc32(yd.to_host(0, cnt).tobytes())
    path = _ffi.debug_path()
    every = int(os.environ.get('SOAK_CHECK_EVERY', '4'))
    t0 = time.time(); diff = 0
    for k in range(reps):
        fn()
        if k % every == every - 1 or k == reps - 1:   # (back-to-back launches between the checks: what a stream of calls looks like)
            _ffi.sync()
            diff += zlib.crc32(yd.to_host(0, cnt).tobytes()) != first
    bad += diff
    print("%-64s %4d launches, %d differing   %s   %.1f s" % (name, reps, diff, path, time.time() - t0), flush=True)
print("soak:", "ALL IDENTICAL" if bad == 0 else "%d DIFFERENCES" % bad)
sys.exit(1 if bad else 0)
