"""Alternating A/B of one integer option on the 127-tap float32 filter (2^26 samples): python tools/ab_fir127.py <option> <v0> <v1> [...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
steps, n = 300, 1 << 26
_ffi.init(0)
xd = _ffi.DeviceArray(n, np.float32).fill_noise(7)
yd = _ffi.DeviceArray(n, np.float32)
k = _ffi.FirKernel(bench.firwin_lowpass(127, 0.2), _ffi.F32)
k12 = _ffi.FirKernel(bench.firwin_lowpass(512, 0.9 / 12), _ffi.F32)
x12 = _ffi.DeviceArray(n // 12, np.float32).fill_noise(9)
def timed(fn):
    for _ in range(150): fn()
    _ffi.sync(); _ffi.timer_start()
    for _ in range(steps): fn()
    return _ffi.timer_stop() / steps
for cname, fn in (("fir127 f32", lambda: k.filter_dev(xd, yd)), ("up12 512 taps f32", lambda: k12.up_dev(x12, yd, 12)), ("dn12 512 taps f32", lambda: k12.dn_dev(xd, yd, 12))):
    out = []
    for rnd in range(3):
        for v in vals:
            _ffi.set_option(name, v)
            out.append("%s=%d %.4f" % (name, v, timed(fn)))
    print(cname, " | ".join(out), flush=True)
