import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from scipy import signal
from sk_dsp_comm_amd import _ffi
_ffi.init(0)
sos8 = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]
n = 1 << 26
for name, sos in (("ellip8", sos8), ("deci24 stage 1: butter(10, 1/2)", signal.butter(10, 0.5, output="sos")), ("cheby1-12", signal.cheby1(12, 0.5, 0.45, output="sos"))):
    for dt in (np.float32, np.complex64):
        row = "%-32s M=2 %-10s" % (name, np.dtype(dt).name)
        for v in (3, 1):
            _ffi.set_option("iir_dn_t96", v)
            k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
            xd = _ffi.DeviceArray(n, dt).fill_noise(5); yd = _ffi.DeviceArray(n // 2 + 16, dt)
            for _ in range(10): k.dn_dev(xd, yd, 2)
            _ffi.sync(); _ffi.timer_start()
            for _ in range(30): k.dn_dev(xd, yd, 2)
            ms = _ffi.timer_stop() / 30
            xd.free(); yd.free()
            rng = np.random.default_rng(3)
            errs = []
            for m in (200_003, 6144 * 3, 6144 * 3 + 1, 97, 5):
                x = rng.standard_normal(m).astype(dt) if np.dtype(dt).kind != "c" else (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(dt)
                got = k.dn(x, 2)
                ref = signal.sosfilt(sos, x.astype(np.complex128 if np.dtype(dt).kind == "c" else np.float64))[::2][:m // 2]
                assert got.shape == ref.shape, (got.shape, ref.shape)
                errs.append(np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-30))
            row += "  %s %.4f ms err %.1e" % ("ranges" if v == 3 else "image ", ms, max(errs))
        print(row, flush=True)
