"""A/B of the IIR scan paths on interleaved complex signals: python tools/ab_iir_c.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from scipy import signal
from sk_dsp_comm_amd import _ffi
def timeit(step, k=100):
    for _ in range(50): step()
    _ffi.sync(); _ffi.timer_start()
    for _ in range(k): step()
    return _ffi.timer_stop() / k
n = 1 << 26
_ffi.init(0)
filters = {"ellip bandpass 8 biquads (config 4)": np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"],
           "butter 8 wn 0.075": signal.butter(8, 0.075, output="sos"), "cheby1 6 wn 0.2": signal.cheby1(6, 0.05, 0.2, output="sos"),
           "butter 4 wn 0.25": signal.butter(4, 0.25, output="sos")}
for name, sos in filters.items():
    for dt in (np.complex64, np.complex128):
        xd = _ffi.DeviceArray(n, dt).fill_noise(7); y1 = _ffi.DeviceArray(n, dt); y2 = _ffi.DeviceArray(n, dt)
        k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
        with _ffi.option("iir_two_pass", -1):
            t1 = timeit(lambda: k.filter_dev(xd, y1))
        with _ffi.option("iir_two_pass", 1):
            t2 = timeit(lambda: k.filter_dev(xd, y2))
        a = y1.to_host(n - (1 << 20), 1 << 20); b = y2.to_host(n - (1 << 20), 1 << 20)
        m = 200000
        ref = signal.sosfilt(sos, xd.to_host(0, m).astype(np.complex128))
        e0 = float(np.max(np.abs(y1.to_host(0, m) - ref)) / np.max(np.abs(ref)))
        print("%-40s %-10s single-pass %.4f ms (%.2f TB/s)  two-pass %.4f ms  diff %.1e  vs sosfilt %.1e" % (
            name, np.dtype(dt).name, t1, 2 * np.dtype(dt).itemsize * n / t1 / 1e9, t2, float(np.max(np.abs(a - b)) / np.max(np.abs(b))), e0), flush=True)
        for d in (xd, y1, y2): d.free()
