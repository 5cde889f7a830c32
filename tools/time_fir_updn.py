"""Developer timing: L / M rate change with long phases -- polyphase kernels, the overlap-save walk with a strided copy of its
full-rate result, the walk whose store keeps every M-th output, and the default dispatch; 2^26 up-rate samples.
Run on the GPU box: python tools/time_fir_updn.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
_ffi.init(0)
for dt in (np.complex64, np.float32, np.float64, np.complex128):
    for L, M, ntaps in ((4, 3, 2048), (4, 3, 4096), (3, 2, 1536), (2, 3, 1024), (12, 5, 6144)):
        n = (1 << 26) // L
        k = _ffi.FirKernel(bench.firwin_lowpass(ntaps, 0.8 / max(L, M)), _ffi.code_of(dt))
        xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n * L // M + 8, dt)
        ms = []
        for thr, fused in ((0, 1), (-2, 0), (-2, 1), (64, 1)):
            with _ffi.option("fir_up_ols_min", thr), _ffi.option("fir_updn_fused", fused):
                for _ in range(3): k.updn_dev(xd, yd, L, M)
                _ffi.sync(); _ffi.timer_start()
                for _ in range(10): k.updn_dev(xd, yd, L, M)
                ms.append(_ffi.timer_stop() / 10)
        print("%-10s L/M=%2d/%d %5d taps n_in %9d: polyphase %.4f  walk+copy %.4f  walk fused %.4f  default %.4f ms" % (np.dtype(dt).name, L, M, ntaps, n, *ms), flush=True)
        xd.free(); yd.free()
