"""Alternating A/B of one integer option on the config-4 kernels (8-biquad elliptic band-pass, 2^26 float32): python tools/ab_option.py <option> <v0> <v1> [...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from sk_dsp_comm_amd import _ffi
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
steps, n = 200, 1 << 26
sos = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]
_ffi.init(0)
k = _ffi.IirKernel(_ffi.F32, sos=sos)
xd = _ffi.DeviceArray(n, np.float32).fill_noise(7)
yd = _ffi.DeviceArray(n, np.float32)
half = _ffi.DeviceArray(n // 2, np.float32).fill_noise(9)
def timed(fn):
    for _ in range(100): fn()
    _ffi.sync(); _ffi.timer_start()
    for _ in range(steps): fn()
    return _ffi.timer_stop() / steps
for cname, fn in (("filter", lambda: k.filter_dev(xd, yd)), ("dn3", lambda: k.dn_dev(xd, yd, 3)), ("up2", lambda: k.up_dev(half, yd, 2))):
    out = []
    for rnd in range(2):
        for v in vals:
            _ffi.set_option(name, v)
            out.append("%s=%d %.4f" % (name, v, timed(fn)))
    print(cname, " | ".join(out), flush=True)
