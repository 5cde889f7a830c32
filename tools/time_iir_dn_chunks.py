"""IIR .dn: 128-sample chunks vs 96-sample chunks (lean compact store from M = 4), float32 / complex64, with parity against the oracle on a short signal"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from scipy import signal
from sk_dsp_comm_amd import _ffi
_ffi.init(0)
sos8 = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]
n = 1 << 26
cases = [("butter8 rc%d" % M, signal.butter(8, 0.9 / M, output="sos"), M) for M in (3, 4, 6, 8, 12, 16, 24, 48, 96)] + [("ellip8", sos8, M) for M in (3, 4, 6, 8, 12)] + [("butter10 (5 biquads)", signal.butter(10, 0.9 / M, output="sos"), M) for M in (4, 8, 16)] + [("cheby1-12 (6 biquads)", signal.cheby1(12, 0.5, 0.9 / M, output="sos"), M) for M in (4, 8)] + [("ellip7 (4 biquads, one real pole)", signal.ellip(7, 0.5, 60, 0.9 / M, output="sos"), M) for M in (4,)] + [("butter14 (7 biquads)", signal.butter(14, 0.9 / M, output="sos"), M) for M in (4, 8)]
for name, sos, M in cases:
    for dt in ((np.float64, np.complex128) if os.environ.get('TIME_F64') else (np.float32, np.complex64)):
        row = "%-14s M=%2d %-10s" % (name, M, np.dtype(dt).name)
        for v in (0, 2):
            _ffi.set_option("iir_dn_t96", v)
            k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
            xd = _ffi.DeviceArray(n, dt).fill_noise(5); yd = _ffi.DeviceArray(n // M + 16, dt)
            for _ in range(10): k.dn_dev(xd, yd, M)
            _ffi.sync(); _ffi.timer_start()
            for _ in range(30): k.dn_dev(xd, yd, M)
            ms = _ffi.timer_stop() / 30
            xd.free(); yd.free()
            # parity, ragged length
            rng = np.random.default_rng(3); m = 200_003
            x = rng.standard_normal(m).astype(dt) if np.dtype(dt).kind != "c" else (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(dt)
            got = k.dn(x, M)
            ref = signal.sosfilt(sos, x.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64))[::M]
            err = np.max(np.abs(got - ref[:got.size])) / np.max(np.abs(ref))
            row += "  t96=%d %.4f ms err %.1e n_out %d/%d" % (v, ms, err, got.size, ref.size)
        print(row, flush=True)
