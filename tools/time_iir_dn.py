"""Time multirate_IIR.dn / rate_change.dn on device vectors, 2^26 input samples: python tools/time_iir_dn.py [<option> <value>]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from scipy import signal
from sk_dsp_comm_amd import _ffi
_ffi.init(0)
if len(sys.argv) > 2:
    _ffi.set_option(sys.argv[1], int(sys.argv[2])); print(sys.argv[1], sys.argv[2])
sos8 = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]
n = 1 << 26
for name, sos, M in (("butter8 rc12", signal.butter(8, 0.075, output="sos"), 12), ("ellip8", sos8, 3), ("ellip8", sos8, 2), ("ellip8", sos8, 4), ("butter8 rc2", signal.butter(8, 0.45, output="sos"), 2), ("butter8 rc4", signal.butter(8, 0.225, output="sos"), 4)):
    for dt in (np.float32, np.complex64, np.float64, np.complex128):
        k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
        xd = _ffi.DeviceArray(n, dt).fill_noise(5); yd = _ffi.DeviceArray(n // M + 16, dt)
        for _ in range(10): k.dn_dev(xd, yd, M)
        _ffi.sync(); _ffi.timer_start()
        for _ in range(30): k.dn_dev(xd, yd, M)
        ms = _ffi.timer_stop() / 30
        for _ in range(10): k.filter_dev(xd, xd) if False else None
        print("%-14s M=%2d %-10s dn %.4f ms  (%.2f TB/s of n (1 + 1/M) samples)" % (name, M, np.dtype(dt).name, ms, np.dtype(dt).itemsize * n * (1 + 1.0 / M) / ms / 1e9), flush=True)
        xd.free(); yd.free()
