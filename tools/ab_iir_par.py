"""A/B timing of the parallel-form single-pass scan (iir_par.hip) against the cascade kernels (option iir_par = 0: cascade-form
single pass / two-pass as the policy picks) on one GPU, several filters, 2^log2n samples:
python tools/ab_iir_par.py [log2n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from scipy import signal
from sk_dsp_comm_amd import _ffi

def timeit(step, k=200):
    for _ in range(100): step()
    _ffi.sync()
    _ffi.timer_start()
    for _ in range(k): step()
    return _ffi.timer_stop() / k

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log2n
_ffi.init(0)
filters = {
    "ellip bandpass 8 biquads (config 4)": np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"],
    "butter lowpass order 8, wn 0.075 (rate_change(12))": signal.butter(8, 0.075, output="sos"),
    "butter lowpass order 8, wn 0.45": signal.butter(8, 0.45, output="sos"),
    "cheby1 lowpass order 6, wn 0.2": signal.cheby1(6, 0.05, 0.2, output="sos"),
    "butter lowpass order 4, wn 0.25": signal.butter(4, 0.25, output="sos"),
    "ellip lowpass order 16, wn 0.3": signal.ellip(16, 0.5, 70, 0.3, output="sos"),
    "one biquad (peaking)": signal.tf2sos(*signal.iirpeak(0.1, 30)),
}
for name, sos in filters.items():
    for dt in (np.float32, np.float64, np.complex64, np.complex128):
        xd = _ffi.DeviceArray(n, dt).fill_noise(7)
        y1 = _ffi.DeviceArray(n, dt); y2 = _ffi.DeviceArray(n, dt)
        k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
        with _ffi.option("iir_par", 1):
            t1 = timeit(lambda: k.filter_dev(xd, y1))
        with _ffi.option("iir_par", 0):
            t2 = timeit(lambda: k.filter_dev(xd, y2), 100 if np.dtype(dt).kind == "c" else 200)
        with _ffi.option("iir_par", 1):
            t1b = timeit(lambda: k.filter_dev(xd, y1))
        a = y1.to_host(n - (1 << 20), 1 << 20); b = y2.to_host(n - (1 << 20), 1 << 20)
        err = float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
        bps = 2 * np.dtype(dt).itemsize * n
        print("%-52s %-8s parallel form %.4f / %.4f ms (%.2f TB/s)   cascade kernels %.4f ms   max diff/peak %.1e" % (
            name, np.dtype(dt).name, t1, t1b, bps / min(t1, t1b) / 1e9, t2, err), flush=True)
        for d in (xd, y1, y2): d.free()
