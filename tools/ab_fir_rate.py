"""The overlap-save rate rows (2^26 samples at the high rate, complex64 and float32), one line per run; alternate with SKDSP_LIB=<another build> for a
same-box A/B: python tools/ab_fir_rate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
steps, n = 150, 1 << 26
_ffi.init(0)
out = []
for dt, code in ((np.complex64, _ffi.C64), (np.float32, _ffi.F32)):
    xd = _ffi.DeviceArray(n, dt).fill_noise(7)
    yd = _ffi.DeviceArray(n, dt)
    def timed(fn):
        for _ in range(60): fn()
        _ffi.sync(); _ffi.timer_start()
        for _ in range(steps): fn()
        return _ffi.timer_stop() / steps
    for kind, M, P in (("dn", 4, 1024), ("dn", 2, 1024), ("dn", 8, 2048), ("dn", 12, 1024), ("up", 4, 1024), ("up", 2, 1024), ("up", 8, 2048)):
        k = _ffi.FirKernel(bench.firwin_lowpass(P, 0.8 / M), code)
        k.set_algo(_ffi.FIR_OLS)
        if kind == "dn":
            ms = timed(lambda: k.dn_dev(xd, yd, M))
        else:
            _ffi.set_option("fir_up_rep", 2)
            ms = timed(lambda: k.up_dev(xd, yd, M, n // M))
        out.append("%s%d/%d%s %.4f" % (kind, M, P, "c" if dt == np.complex64 else "r", ms))
    xd.free(); yd.free()
print(os.path.basename(os.environ.get("SKDSP_LIB", "in-tree")), " | ".join(out), _ffi.debug_path(), flush=True)
