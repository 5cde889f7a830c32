"""One FIR call repeated (a rocprofv3 target): python tools/run_fir_call.py <filter|dn|up> <ntaps> <factor> [dtype=c64|f32] [steps=100] [opt=val ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
kind, ntaps, fac = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dt = sys.argv[4] if len(sys.argv) > 4 else "c64"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 100
_ffi.init(0)
for kv in sys.argv[6:]:
    k, v = kv.split("=")
    _ffi.set_option(k, int(v))
npdt, code = (np.complex64, _ffi.C64) if dt == "c64" else (np.float32, _ffi.F32)
n = 1 << 26
xd = _ffi.DeviceArray(n if kind != "up" else n // fac, npdt).fill_noise(7)
yd = _ffi.DeviceArray(n, npdt)
k = _ffi.FirKernel(bench.firwin_lowpass(ntaps, 0.8 / max(fac, 1) if kind != "filter" else 0.2), code)
fn = {"filter": lambda: k.filter_dev(xd, yd), "dn": lambda: k.dn_dev(xd, yd, fac), "up": lambda: k.up_dev(xd, yd, fac)}[kind]
for _ in range(30): fn()
_ffi.sync(); _ffi.timer_start()
for _ in range(steps): fn()
ms = _ffi.timer_stop() / steps
print("%s %d taps by %d %s: %.4f ms  path %s" % (kind, ntaps, fac, dt, ms, _ffi.debug_path()), flush=True)
