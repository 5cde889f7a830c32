"""Time BASELINE config 3 (fused 4/3 resampler, 512 taps, complex64 2^26) for values of one option:
python tools/time_updn43.py <option> v1 v2 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
args = sys.argv[1:]
opt, vals = (args[0], [int(v) for v in args[1:]]) if args else (None, [0])
n = 1 << 26
_ffi.init(0)
b = bench.firwin_lowpass(512, 0.225)
k = _ffi.FirKernel(b, _ffi.C64)
xd = _ffi.DeviceArray(n, np.complex64).fill_noise(1)
yd = _ffi.DeviceArray(n * 4 // 3, np.complex64)
def t(steps=200):
    for _ in range(100): k.updn_dev(xd, yd, 4, 3)
    _ffi.sync(); _ffi.timer_start()
    for _ in range(steps): k.updn_dev(xd, yd, 4, 3)
    return _ffi.timer_stop() / steps
ref = None
for rep in range(2):
    for v in vals:
        if opt: _ffi.set_option(opt, v)
        ms = t()
        y = yd.to_host(5_000_000, 100000)
        if ref is None: ref = y
        print("%s=%s: %.4f ms  %.2f TB/s  same=%s" % (opt, v, ms, (8 * n + 8 * (n * 4 // 3)) / ms / 1e9, bool(np.array_equal(y, ref))), flush=True)
