"""The four matrix-pipe FIR bench rows (fir127, config 3, default .up / .dn) through the library SKDSP_LIB points at, 2^26 samples: python tools/time_bx_rows.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
steps, n = 300, 1 << 26
_ffi.init(0)
xr = _ffi.DeviceArray(n, np.float32).fill_noise(7); yr = _ffi.DeviceArray(n, np.float32)
xc = _ffi.DeviceArray(n, np.complex64).fill_noise(8); yc = _ffi.DeviceArray((n * 4) // 3 + 16, np.complex64)
x12 = _ffi.DeviceArray(n // 12, np.complex64).fill_noise(9)
k127 = _ffi.FirKernel(bench.firwin_lowpass(127, 0.2), _ffi.F32)
k43 = _ffi.FirKernel(bench.firwin_lowpass(512, 0.225), _ffi.C64)
k12 = _ffi.FirKernel(bench.firwin_lowpass(512, 0.9 / 12), _ffi.C64)
def timed(fn):
    for _ in range(150): fn()
    _ffi.sync(); _ffi.timer_start()
    for _ in range(steps): fn()
    return _ffi.timer_stop() / steps
tag = os.path.basename(os.environ.get("SKDSP_LIB", "default"))
print(tag, " ".join("%s %.4f" % (nm, timed(fn)) for nm, fn in (("fir127", lambda: k127.filter_dev(xr, yr)), ("updn43", lambda: k43.updn_dev(xc, yc, 4, 3)),
                                                                ("firup12", lambda: k12.up_dev(x12, yc, 12)), ("firdn12", lambda: k12.dn_dev(xc, yc, 12)))), flush=True)
