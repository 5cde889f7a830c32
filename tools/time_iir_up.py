"""Time multirate_IIR.up / rate_change.up on device vectors (float32, complex64): python tools/time_iir_up.py [<option> <value>]
   algorithmic bytes = 4 B x (n_in + n_in L): the input read once, the output written once."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from scipy import signal
from sk_dsp_comm_amd import _ffi
from oracle import oracle as orc
_ffi.init(0)
if len(sys.argv) > 2:
    _ffi.set_option(sys.argv[1], int(sys.argv[2]))
    print("%s = %s" % (sys.argv[1], sys.argv[2]))
sos8 = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]
cases = [("rate_change(12).up: butter(8, 0.075)", signal.butter(8, 0.9 / 12, output="sos"), 12, 1 << 22),
         ("rate_change(4).up: butter(8, 0.225)", signal.butter(8, 0.9 / 4, output="sos"), 4, 1 << 24),
         ("multirate_IIR(ellip bandpass, 8 biquads).up(x, 4)", sos8, 4, 1 << 24),
         ("multirate_IIR(ellip bandpass, 8 biquads).up(x, 2)", sos8, 2, 1 << 25),
         ("multirate_IIR(ellip bandpass, 8 biquads).up(x) [12]", sos8, 12, 1 << 22),
         ("multirate_IIR(ellip lowpass, 6 biquads).up(x) [12]", signal.ellip(12, 0.5, 70, 0.9 / 12, output="sos"), 12, 1 << 22),
         ("multirate_IIR(butter lowpass, 5 biquads).up(x, 8)", signal.butter(10, 0.9 / 8, output="sos"), 8, 1 << 23),
         ("interp24 stage 2: butter(10, 1/3).up(x, 3)", signal.butter(10, 1 / 3., output="sos"), 3, 22369621),
         ("interp24 stage 3: butter(10, 1/4).up(x, 4)", signal.butter(10, 1 / 4., output="sos"), 4, 1 << 24)]
for name, sos, L, n in cases:
    for dt in ((np.float64, np.complex128) if os.environ.get("TIME_F64") else (np.float32, np.complex64)):
        k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
        xd = _ffi.DeviceArray(n, dt).fill_noise(5); yd = _ffi.DeviceArray(n * L, dt)
        for _ in range(20): k.up_dev(xd, yd, L)
        _ffi.sync(); _ffi.timer_start()
        for _ in range(50): k.up_dev(xd, yd, L)
        ms = _ffi.timer_stop() / 50
        x = xd.to_host(0, 40000)
        up = np.zeros(40000 * L, dtype=np.complex128 if np.dtype(dt).kind == 'c' else np.float64); up[::L] = L * x.astype(up.dtype)
        ref = signal.sosfilt(sos, up)
        e = float(np.max(np.abs(yd.to_host(0, 40000 * L) - ref)) / np.max(np.abs(ref)))
        print("%-52s %s n_in 2^%d: %.4f ms  %.2f TB/s algorithmic  err %.1e" % (name, np.dtype(dt).name, n.bit_length() - 1, ms, np.dtype(dt).itemsize * n * (1 + L) / ms / 1e9, e), flush=True)
        xd.free(); yd.free()
