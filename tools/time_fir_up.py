"""Developer timing: multirate_FIR.up through its engines -- the polyphase kernels, the overlap-save walk over (tile, phase) pairs (strided
stores / rows + weave), the one-workgroup-per-input-tile interpolators (fir_up4k / fir_up2k) -- and what the default dispatch takes,
device-resident signals, 2^NOUT_LOG2 outputs (default 26), settled clock.
Run on the GPU box: python tools/time_fir_up.py [LxT ...]"""
import time
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi

_ffi.init(0)
shapes = [(L, T) for L in (2, 3, 4, 8, 12) for T in (48, 64, 96, 128, 192, 256, 512, 1024)] + [(5, 256), (11, 256), (64, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
DTYPES = [np.dtype(d).type for d in os.environ.get("DTYPES", "complex64,float32").split(",")]
for dt in DTYPES:
    for L, T in shapes:
        ntaps = L * T
        n = (1 << int(os.environ.get("NOUT_LOG2", "26"))) // L
        k = _ffi.FirKernel(bench.firwin_lowpass(ntaps, 0.8 / L), _ffi.code_of(dt))
        xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n * L, dt)
        ms = []
        for thr, rows, tile, rep in ((0, 0, 0, 0), (-2, 0, 0, 0), (-2, 2, 0, 0), (-2, -1, 2, 0), (64, -1, 1, 2), (64, -1, 1, 1)):
            if rep == 2 and (L % 2 or ntaps > 4097):
                ms.append(float("nan"))
                continue
            with _ffi.option("fir_up_ols_min", thr), _ffi.option("fir_up_rows_min", rows), _ffi.option("fir_up4k", tile), _ffi.option("fir_up_rep", rep):
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.15:
                    for _ in range(10): k.up_dev(xd, yd, L)
                    _ffi.sync()
                _ffi.timer_start()
                for _ in range(40): k.up_dev(xd, yd, L)
                ms.append(_ffi.timer_stop() / 40)
        isz = np.dtype(dt).itemsize
        best = np.nanmin(ms[:5])
        print("%-10s up L=%2d %5d taps (%4d per phase) n_in %9d: polyphase %.4f ms  walk, strided stores %.4f  walk, rows + weave %.4f  input-tile interpolator %.4f  replicated spectrum %.4f  default %.4f ms (%.2f TB/s algorithmic)%s"
              % (np.dtype(dt).name, L, ntaps, T, n, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], isz * n * (1 + L) / ms[5] / 1e9,
                 "" if ms[5] <= 1.08 * best else "   <-- default is not the fastest path"), flush=True)
        xd.free(); yd.free()
