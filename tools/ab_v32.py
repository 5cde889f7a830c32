"""A/B of the float32 from-rest states (option iir_par_v32) on the 8-biquad elliptic band-pass of BASELINE config 4, 2^26 float32 samples:
.filter, .dn(x, 3), .up(x, 2) -- alternating timings on one box, and the error of each form against the CPU oracle on noise, DC, the Nyquist
alternation and a tone on every section's resonance (the inputs the plan's probe admits the filter on).   python tools/ab_v32.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from sk_dsp_comm_amd import _ffi
from oracle import oracle as orc
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = 1 << 26
sos = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]
_ffi.init(0)
info = _ffi.sos_par_info(sos)
print("kappa %.2f  probe error T=128 %.2e  T=96 %.2e  admitted %s" % (info["kappa"], info["v32_err"], info["v32_err_t96"], info["v32_admitted"]))
k = _ffi.IirKernel(_ffi.F32, sos=sos)
xd = _ffi.DeviceArray(n, np.float32).fill_noise(7)
yd = _ffi.DeviceArray(n, np.float32)
def timed(fn):
    for _ in range(max(steps // 2, 20)): fn()
    _ffi.sync(); _ffi.timer_start()
    for _ in range(steps): fn()
    return _ffi.timer_stop() / steps
calls = {"filter": lambda: k.filter_dev(xd, yd), "dn3": lambda: k.dn_dev(xd, yd, 3), "up2": lambda: k.up_dev(_half, yd, 2)}
_half = _ffi.DeviceArray(n // 2, np.float32).fill_noise(9)
for name, fn in calls.items():
    row = []
    for rnd in range(3):
        for v in (0, 1):
            _ffi.set_option("iir_par_v32", v)
            _ffi.debug_path()
            t = timed(fn)
            row.append((v, t, "iir_par_v32" in _ffi.debug_path()))
    print(name, "  ".join("v32=%d%s %.4f ms" % (v, "*" if used else "", t) for v, t, used in row), flush=True)
# accuracy: head of the result on the probe inputs
m = 1 << 17
t = np.arange(m)
rng = np.random.default_rng(3)
probes = {"noise": rng.standard_normal(m), "dc": np.ones(m), "nyquist": (-1.0) ** t}
for i, (a1, a2, r0, r1) in enumerate(info["sections"]):
    if a2 > 0 and a1 * a1 < 4 * a2:
        probes["res%d" % i] = np.cos(np.arccos(-a1 / (2 * np.sqrt(a2))) * t)
for v in (0, 1):
    _ffi.set_option("iir_par_v32", v)
    worst = {}
    for nm, x in probes.items():
        x32 = x.astype(np.float32)
        xs = _ffi.DeviceArray.from_host(x32); ys = _ffi.DeviceArray(m, np.float32)
        k.filter_dev(xs, ys)
        ref = orc.sos_filter(sos, x32)
        worst[nm] = float(np.max(np.abs(ys.to_host() - ref)) / np.max(np.abs(ref)))
        y3 = _ffi.DeviceArray(m // 3, np.float32)
        k.dn_dev(xs, y3, 3)
        worst[nm + "/dn3"] = float(np.max(np.abs(y3.to_host() - ref[:(m // 3) * 3:3])) / np.max(np.abs(ref)))
        xs.free(); ys.free(); y3.free()
    print("v32=%d  worst %.2e   " % (v, max(worst.values())) + " ".join("%s %.1e" % kv for kv in worst.items()), flush=True)
