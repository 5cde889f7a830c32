"""Developer check + timing of the frequency-domain decimator (fir_dn4k.hip) against the oracle and the older engines.
Run on the GPU box: python tools/check_dn4k.py [check] [time]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
from oracle import oracle as orc

_ffi.init(0)
what = sys.argv[1:] or ["check", "time"]
rng = np.random.default_rng(4)
if "check" in what:
    worst = 0.0
    for dt in (np.complex64, np.float32):
        for M, ntaps, n in ((4, 1024, 160000), (3, 512, 100001), (2, 3000, 120011), (3, 700, 40000), (4, 64, 50000), (2, 4097, 200000), (4, 1024, 4 * 3840), (4, 1024, 4 * 3841 + 3), (3, 777, 3 * 8191)):
            for cplx_taps in ((False, True) if dt == np.complex64 else (False,)):
                b = rng.standard_normal(ntaps) / np.sqrt(ntaps)
                if cplx_taps:
                    b = b + 1j * rng.standard_normal(ntaps) / np.sqrt(ntaps)
                x = rng.standard_normal(n).astype(np.float32)
                if dt == np.complex64:
                    x = (x + 1j * rng.standard_normal(n)).astype(np.complex64)
                k = _ffi.FirKernel(b, _ffi.code_of(dt))
                ref = orc.fir_dn(b, x, M)
                with _ffi.option("fir_dn4k", 2):
                    xd = _ffi.DeviceArray.from_host(x); yd = _ffi.DeviceArray(n // M + 64, dt)
                    _ffi.check(_ffi.load().skdsp_memset(__import__("ctypes").c_void_p(yd.ptr), 0x7f, (n // M + 64) * np.dtype(dt).itemsize))
                    k.dn_dev(xd, yd, M, n)
                    got = yd.to_host(0, n // M)
                    guard = yd.to_host(n // M, 64)
                e = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
                clean = bool(np.all(guard.view(np.uint8) == 0x7f))
                worst = max(worst, e)
                print("check %-9s M=%d %4d taps%s n=%6d: err %.2e guard %s%s" % (np.dtype(dt).name, M, ntaps, " (complex)" if cplx_taps else "", n, e, clean, "" if (e < 1e-6 and clean) else "   <-- FAIL"), flush=True)
                xd.free(); yd.free()
    for dt in (np.complex64, np.float32):   # a streamed continuation
        M, ntaps, n = 4, 1024, 160000
        b = rng.standard_normal(ntaps) / 32
        x = rng.standard_normal(n).astype(np.float32)
        if dt == np.complex64:
            x = (x + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = orc.fir_dn(b, x, M)
        k = _ffi.FirKernel(b, _ffi.code_of(dt))
        with _ffi.option("fir_dn4k", 2):
            xd = _ffi.DeviceArray.from_host(x); yd = _ffi.DeviceArray(n // M, dt)
            half = (n // 2 // M) * M
            k.dn_dev(xd, yd, M, half)
            k.dn_dev(xd.window(half, n - half), yd.window(half // M, (n - half) // M), M, n - half, n_hist=ntaps - 1)
            got = yd.to_host()
        e = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
        print("check %-9s streamed continuation (n_hist = Ntaps - 1): err %.2e%s" % (np.dtype(dt).name, e, "" if e < 1e-6 else "   <-- FAIL"))
        worst = max(worst, e)
    print("check worst %.2e" % worst)

if "time" in what:
    for dt in (np.complex64, np.float32):
        for M, ntaps in ((3, 512), (4, 1024), (2, 1024), (4, 256), (3, 128), (4, 4096)):
            n = 1 << 26
            k = _ffi.FirKernel(bench.firwin_lowpass(ntaps, 0.8 / M), _ffi.code_of(dt))
            xd = _ffi.DeviceArray(n, dt).fill_noise(1); yd = _ffi.DeviceArray(n // M, dt)
            ms = []
            for v in (0, 2):
                with _ffi.option("fir_dn4k", v):
                    t0 = time.perf_counter()
                    while time.perf_counter() - t0 < 0.25:
                        for _ in range(20): k.dn_dev(xd, yd, M)
                        _ffi.sync()
                    _ffi.timer_start()
                    for _ in range(100): k.dn_dev(xd, yd, M)
                    ms.append(_ffi.timer_stop() / 100)
            isz = np.dtype(dt).itemsize
            print("time %-9s dn M=%d %5d taps 2^26 inputs: older engines %.4f ms | frequency-domain decimator %.4f ms (%.2f TB/s algorithmic, %.1f %% of 8 TB/s)"
                  % (np.dtype(dt).name, M, ntaps, ms[0], ms[1], isz * (n + n // M) / ms[1] / 1e9, isz * (n + n // M) / ms[1] / 1e9 / 80), flush=True)
            xd.free(); yd.free()
