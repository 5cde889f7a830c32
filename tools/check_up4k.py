"""Developer check + timing of the one-workgroup-per-input-tile interpolator (fir_up4k.hip) against the oracle and the older engines.
Run on the GPU box: python tools/check_up4k.py [check] [time]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
from oracle import oracle as orc

_ffi.init(0)
what = sys.argv[1:] or ["check", "time"]
rng = np.random.default_rng(3)
if "check" in what:
    worst = 0.0
    for dt in (np.complex64, np.float32):
        for L, ntaps, n in ((4, 1024, 40000), (12, 512, 9000), (2, 3000, 30011), (3, 700, 12345), (5, 777, 20000), (7, 64, 5000), (8, 2048, 16384), (16, 333, 4097),
                            (2, 4097, 50000), (4, 1024, 3840), (4, 1024, 3841), (13, 1300, 8191)):
            for cplx_taps in ((False, True) if dt == np.complex64 else (False,)):
                b = rng.standard_normal(ntaps) / np.sqrt(ntaps)
                if cplx_taps:
                    b = b + 1j * rng.standard_normal(ntaps) / np.sqrt(ntaps)
                x = rng.standard_normal(n).astype(np.float32)
                if dt == np.complex64:
                    x = (x + 1j * rng.standard_normal(n)).astype(np.complex64)
                k = _ffi.FirKernel(b, _ffi.code_of(dt))
                ref = orc.fir_up(b, x, L)
                for G in (4, 2):
                    with _ffi.option("fir_up4k", 2), _ffi.option("fir_up4k_group", G):
                        xd = _ffi.DeviceArray.from_host(x); yd = _ffi.DeviceArray(n * L + 64, dt)
                        _ffi.check(_ffi.load().skdsp_memset(__import__("ctypes").c_void_p(yd.ptr), 0x7f, (n * L + 64) * np.dtype(dt).itemsize))
                        k.up_dev(xd, yd, L, n)
                        got = yd.to_host(0, n * L)
                        guard = yd.to_host(n * L, 64)
                        e = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
                        clean = bool(np.all(guard.view(np.uint8) == 0x7f))
                        worst = max(worst, e)
                        flag = "" if (e < 1e-6 and clean) else "   <-- FAIL"
                        print("check %-9s L=%2d %4d taps%s n=%6d G=%d: err %.2e guard %s%s" % (np.dtype(dt).name, L, ntaps, " (complex)" if cplx_taps else "", n, G, e, clean, flag), flush=True)
                        xd.free(); yd.free()
    # a streamed continuation: the second half of a signal with the first half's tail as history
    for dt in (np.complex64, np.float32):
        L, ntaps, n = 4, 1024, 30000
        b = rng.standard_normal(ntaps) / 32
        x = rng.standard_normal(n).astype(np.float32)
        if dt == np.complex64:
            x = (x + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = orc.fir_up(b, x, L)
        k = _ffi.FirKernel(b, _ffi.code_of(dt))
        h = 300
        with _ffi.option("fir_up4k", 2):
            xd = _ffi.DeviceArray.from_host(x); yd = _ffi.DeviceArray(n * L, dt)
            half = n // 2
            k.up_dev(xd, yd, L, half)
            k.up_dev(xd.window(half, n - half), yd.window(half * L, (n - half) * L), L, n - half, n_hist=h)
            got = yd.to_host()
        e = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
        print("check %-9s streamed continuation (n_hist = %d >= taps per phase - 1): err %.2e%s" % (np.dtype(dt).name, h, e, "" if e < 1e-6 else "   <-- FAIL"))
        worst = max(worst, e)
    print("check worst %.2e" % worst)

if "time" in what:
    shapes = [(2, 512), (4, 256), (4, 64), (4, 1024), (8, 128), (12, 43), (12, 256), (3, 256), (5, 256), (6, 128), (64, 64)]
    for dt in (np.complex64, np.float32):
        for L, T in shapes:
            ntaps = L * T
            n = (1 << 26) // L
            k = _ffi.FirKernel(bench.firwin_lowpass(ntaps, 0.8 / L), _ffi.code_of(dt))
            xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n * L, dt)
            ms = []
            for opts in ((("fir_up4k", 0),), (("fir_up4k", 2), ("fir_up4k_group", 4)), (("fir_up4k", 2), ("fir_up4k_group", 2))):
                ctxs = [_ffi.option(a, b) for a, b in opts]
                for c in ctxs: c.__enter__()
                for _ in range(5): k.up_dev(xd, yd, L)
                _ffi.sync(); _ffi.timer_start()
                for _ in range(20): k.up_dev(xd, yd, L)
                ms.append(_ffi.timer_stop() / 20)
                for c in reversed(ctxs): c.__exit__(None, None, None)
            isz = np.dtype(dt).itemsize
            print("time %-9s up L=%2d %5d taps (%4d per phase) 2^26 outputs: older engines %.4f ms | tile kernel, 4 phases per store %.4f (%.2f TB/s algorithmic) | 2 phases %.4f"
                  % (np.dtype(dt).name, L, ntaps, T, ms[0], ms[1], isz * n * (1 + L) / ms[1] / 1e9, ms[2]), flush=True)
            xd.free(); yd.free()
