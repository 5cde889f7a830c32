"""Developer timing: multirate_FIR.dn through its engines -- the polyphase kernels (kept outputs only), the overlap-save tile with the
decimating store, the frequency-domain decimator (fir_dn4k, M <= 4) -- and what the default dispatch takes; device-resident signals,
2^26 inputs, settled clock.
Run on the GPU box: python tools/time_fir_dn.py [MxNTAPS ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi

_ffi.init(0)
shapes = [(M, P) for M in (2, 3, 4, 8, 12, 16, 24) for P in (128, 256, 512, 1024, 2048)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
DTYPES = [np.dtype(d).type for d in os.environ.get("DTYPES", "complex64,float32").split(",")]
n = 1 << int(os.environ.get("NIN_LOG2", "26"))
for dt in DTYPES:
    for M, P in shapes:
        xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n // M, dt)
        ms = []
        for algo, dn4k, fold in ((_ffi.FIR_DIRECT, 0, 0), (_ffi.FIR_OLS, 0, 0), (_ffi.FIR_OLS, 0, 1), (_ffi.FIR_OLS, 2, 1), (None, 1, 1)):
            k = _ffi.FirKernel(bench.firwin_lowpass(P, 0.8 / M), _ffi.code_of(dt))
            if algo is not None:
                k.set_algo(algo)
            if (dn4k == 2 and M > 4) or (algo == _ffi.FIR_OLS and dn4k == 0 and fold == 1 and M % 2):
                ms.append(float("nan"))
                continue
            try:
                with _ffi.option("fir_dn4k", dn4k), _ffi.option("fir_dn_fold", fold):
                    t0 = time.perf_counter()
                    while time.perf_counter() - t0 < 0.15:
                        for _ in range(10): k.dn_dev(xd, yd, M)
                        _ffi.sync()
                    _ffi.timer_start()
                    for _ in range(40): k.dn_dev(xd, yd, M)
                    ms.append(_ffi.timer_stop() / 40)
            except Exception:
                ms.append(float("nan"))
        isz = np.dtype(dt).itemsize
        best = np.nanmin(ms[:4])
        print("%-10s dn M=%2d %5d taps: polyphase %.4f ms  overlap-save, decimating store %.4f  folded inverse %.4f  frequency-domain decimator %.4f  default %.4f ms (%.2f TB/s algorithmic)%s"
              % (np.dtype(dt).name, M, P, ms[0], ms[1], ms[2], ms[3], ms[4], isz * (n + n // M) / ms[4] / 1e9,
                 "" if ms[4] <= 1.08 * best else "   <-- default is not the fastest path"), flush=True)
        xd.free(); yd.free()
