"""Developer timing: multirate_FIR.filter -- the direct form (matrix-pipe kernel where it covers the shape) against overlap-save and the
default dispatch, device-resident signals, 2^26 samples, settled clock: where the crossover of pick_fir_algo (capi.hip) comes from.
Run on the GPU box: python tools/time_fir_filter.py [NTAPS ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi

_ffi.init(0)
taps = [int(a) for a in sys.argv[1:]] or [32, 64, 81, 96, 128, 145, 160, 192, 224, 256, 288, 320, 384, 448, 512]
n = 1 << 26
for dt in (np.complex64, np.float32):
    xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n, dt)
    for P in taps:
        ms = []
        for algo in (_ffi.FIR_DIRECT, _ffi.FIR_OLS, None):
            k = _ffi.FirKernel(bench.firwin_lowpass(P, 0.2), _ffi.code_of(dt))
            if algo is not None:
                k.set_algo(algo)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.15:
                for _ in range(10): k.filter_dev(xd, yd)
                _ffi.sync()
            _ffi.timer_start()
            for _ in range(40): k.filter_dev(xd, yd)
            ms.append(_ffi.timer_stop() / 40)
        isz = np.dtype(dt).itemsize
        print("%-10s filter %4d taps: direct %.4f ms  overlap-save %.4f  default %.4f ms (%.2f TB/s algorithmic)%s"
              % (np.dtype(dt).name, P, ms[0], ms[1], ms[2], 2 * isz * n / ms[2] / 1e9, "" if ms[2] <= 1.05 * min(ms[:2]) else "   <-- default is not the fastest path"), flush=True)
    xd.free(); yd.free()
