"""Time the headline overlap-save kernel (1024 taps, complex64, 2^log2n) for several values of one option:
python tools/time_fir1024.py <option> v1 v2 ... [--log2n 26]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
args = sys.argv[1:]
log2n = 26
if "--log2n" in args:
    i = args.index("--log2n"); log2n = int(args[i + 1]); del args[i:i + 2]
opt, vals = (args[0], [int(v) for v in args[1:]]) if args else (None, [0])
n = 1 << log2n
_ffi.init(0)
b = bench.firwin_lowpass(1024, 0.2)
k = _ffi.FirKernel(b, _ffi.C64)
xd = _ffi.DeviceArray(n, np.complex64, headroom=1024).fill_noise(1)
yd = _ffi.DeviceArray(n, np.complex64)
def t(steps=300):
    for _ in range(150): k.filter_dev(xd, yd)
    _ffi.sync(); _ffi.timer_start()
    for _ in range(steps): k.filter_dev(xd, yd)
    return _ffi.timer_stop() / steps
for rep in range(2):
    for v in vals:
        if opt: _ffi.set_option(opt, v)
        ms = t()
        print("%s %s=%s: %.4f ms  %.1f %% of 8 TB/s" % (os.path.basename(os.environ.get("SKDSP_LIB", "default")), opt, v, ms, 16 * n / ms / 1e9 / 80), flush=True)
