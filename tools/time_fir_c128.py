"""Time the float64 overlap-save FIR (1024 taps): python tools/time_fir_c128.py [<option> v1 v2 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
from oracle import oracle as orc
_ffi.init(0)
b = bench.firwin_lowpass(1024, 0.2)
args = sys.argv[1:]
opt, vals = (args[0], [int(v) for v in args[1:]]) if args else (None, [None])
for val, dt, n in [(v, d, 1 << 26) for v in vals for d in (np.complex128, np.float64)]:
    if opt:
        _ffi.set_option(opt, val)
    k = _ffi.FirKernel(b, _ffi.code_of(dt)); k.set_algo(_ffi.FIR_OLS)
    xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n, dt)
    for _ in range(20): k.filter_dev(xd, yd)
    _ffi.sync(); _ffi.timer_start()
    for _ in range(50): k.filter_dev(xd, yd)
    t = _ffi.timer_stop() / 50
    s0 = 3072 * 1000 - 77
    ref = orc.fir_filter(b, xd.to_host(s0 - 1023, 5000 + 1023))[1023:]
    e = float(np.max(np.abs(yd.to_host(s0, 5000) - ref)) / np.max(np.abs(ref)))
    print("%s %s%s 2^26 1024 taps: %.4f ms  %.2f TB/s  err %.1e" % (os.path.basename(os.environ.get("SKDSP_LIB", "default")), "%s=%s " % (opt, val) if opt else "", np.dtype(dt).name, t, 2 * np.dtype(dt).itemsize * n / t / 1e9, e), flush=True)
    xd.free(); yd.free()
