"""Time the 8-biquad float32 IIR (2^26) through the library SKDSP_LIB points at: python tools/time_iir8.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from sk_dsp_comm_amd import _ffi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 1 << 26
sos = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]
_ffi.init(0)
xd = _ffi.DeviceArray(n, np.float32).fill_noise(7)
yd = _ffi.DeviceArray(n, np.float32)
k = _ffi.IirKernel(_ffi.F32, sos=sos)
for _ in range(max(steps // 2, 20)): k.filter_dev(xd, yd)
_ffi.sync(); _ffi.timer_start()
for _ in range(steps): k.filter_dev(xd, yd)
t = _ffi.timer_stop() / steps
print("%s: %.4f ms  %.2f TB/s" % (os.path.basename(os.environ.get("SKDSP_LIB", "default")), t, 8 * n / t / 1e9), flush=True)
