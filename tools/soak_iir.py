"""Launch-to-launch determinism of the parallel-form IIR scan and its rate forms (the look-back sums from-rest states of predecessors in a fixed order: every launch must give the same bits).
Run on the GPU box: python tools/soak_iir.py"""
import os, sys, zlib
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi
_ffi.init(0)
n = 1 << 24
xr = _ffi.DeviceArray(n, np.float32).fill_noise(4); yr = _ffi.DeviceArray(n, np.float32)
xc = _ffi.DeviceArray(n, np.complex64).fill_noise(5); yc = _ffi.DeviceArray(n, np.complex64)
xd = _ffi.DeviceArray(n, np.float64).fill_noise(6); yd = _ffi.DeviceArray(n, np.float64)
sos8 = bench.elliptic_bpf_sos()
sos4 = _ffi.tf2sos(*bench._butter8_rate_change12())
k8 = _ffi.IirKernel(_ffi.F32, sos=sos8); k8c = _ffi.IirKernel(_ffi.C64, sos=sos8); k8d = _ffi.IirKernel(_ffi.F64, sos=sos8); k4 = _ffi.IirKernel(_ffi.F32, sos=sos4)
cases = [("iir8 f32", lambda: k8.filter_dev(xr, yr), yr, n), ("iir8 c64", lambda: k8c.filter_dev(xc, yc), yc, n), ("iir8 f64", lambda: k8d.filter_dev(xd, yd), yd, n),
         ("iirlp8 f32", lambda: k4.filter_dev(xr, yr), yr, n), ("iir8 dn3", lambda: k8.dn_dev(xr, yr, 3), yr, n // 3), ("iir8 up2", lambda: k8.up_dev(xr, yr, 2, n // 2), yr, n),
         ("rcup12", lambda: k4.up_dev(xr, yr, 12, n // 12), yr, (n // 12) * 12), ("rcdn12", lambda: k4.dn_dev(xr, yr, 12), yr, n // 12)]
for name, fn, y, cnt in cases:
    _ffi.debug_path(); fn(); _ffi.sync()
    first = y.to_host(0, cnt).copy(); path = _ffi.debug_path()
    ndiff = 0; worst = 0.0
    for k in range(100):
        fn()
        if k % 4 == 3:
            _ffi.sync(); got = y.to_host(0, cnt)
            if not np.array_equal(got, first):
                ndiff += 1; worst = max(worst, float(np.max(np.abs(got - first)) / np.max(np.abs(first))))
    print("%-12s %s: %d of 25 checks differ, worst %.2e of the peak" % (name, path, ndiff, worst), flush=True)
