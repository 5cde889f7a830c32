"""Developer timing of up4k_kernel with a settled clock: python tools/time_up4k.py [dtype] LxT ... (options through SKDSP_<NAME> or `opt=value` args)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
_ffi.init(0)
dt = np.complex64
shapes, variants = [], []
for a in sys.argv[1:]:
    if a in ("complex64", "float32"): dt = np.dtype(a).type
    elif "=" in a: variants.append([(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")])
    else: shapes.append(tuple(int(v) for v in a.split("x")))
variants = variants or [[("fir_up4k", 2)]]
reps = int(os.environ.get("REPS", "200"))
for L, T in shapes:
    n = (1 << 26) // L
    k = _ffi.FirKernel(bench.firwin_lowpass(L * T, 0.8 / L), _ffi.code_of(dt))
    xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n * L, dt)
    for var in variants:
        ctxs = [_ffi.option(a, b) for a, b in var]
        for c in ctxs: c.__enter__()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            for _ in range(20): k.up_dev(xd, yd, L)
            _ffi.sync()
        _ffi.timer_start()
        for _ in range(reps): k.up_dev(xd, yd, L)
        ms = _ffi.timer_stop() / reps
        for c in reversed(ctxs): c.__exit__(None, None, None)
        print("%-9s L=%2d T=%4d %-60s %.4f ms  %.2f TB/s" % (np.dtype(dt).name, L, T, " ".join("%s=%d" % kv for kv in var), ms, np.dtype(dt).itemsize * n * (1 + L) / ms / 1e9), flush=True)
    xd.free(); yd.free()
