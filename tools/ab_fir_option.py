"""Alternating A/B of one integer option on the overlap-save rows (2^26 complex64): python tools/ab_fir_option.py <option> <v0> <v1> [...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
steps, n = 300, 1 << 26
_ffi.init(0)
xd = _ffi.DeviceArray(n, np.complex64).fill_noise(7)
yd = _ffi.DeviceArray(n, np.complex64)
k = _ffi.FirKernel(bench.firwin_lowpass(1024, 0.2), _ffi.C64)
k4 = _ffi.FirKernel(bench.firwin_lowpass(1024, 0.05), _ffi.C64)
x4 = _ffi.DeviceArray(n // 4, np.complex64).fill_noise(9)
def timed(fn):
    for _ in range(150): fn()
    _ffi.sync(); _ffi.timer_start()
    for _ in range(steps): fn()
    return _ffi.timer_stop() / steps
for cname, fn in (("filter1024", lambda: k.filter_dev(xd, yd)), ("dn4", lambda: k4.dn_dev(xd, yd, 4)), ("up4", lambda: k4.up_dev(x4, yd, 4))):
    out = []
    for rnd in range(3):
        for v in vals:
            _ffi.set_option(name, v)
            out.append("%s=%d %.4f" % (name, v, timed(fn)))
    print(cname, " | ".join(out), flush=True)
