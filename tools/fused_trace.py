"""Developer aid: phase timeline of the single-pass IIR kernel (library built with -DSK_FUSED_TRACE_BUILD):
   SKDSP_LIB=.../libskdsp_hip_tr.so python tools/fused_trace.py [out.csv] [sos8 | lp8 | bq1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/fused_trace.csv"
from sk_dsp_comm_amd import _ffi
n = 1 << 26
which = sys.argv[2] if len(sys.argv) > 2 else "sos8"
if which == "sos8":
    sos = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]   # BASELINE config 4
elif which == "lp8":
    from scipy import signal
    sos = signal.butter(8, 0.9 / 12, output="sos")                                  # rate_change(12)'s default design
else:
    from scipy import signal
    sos = signal.tf2sos(*signal.iirpeak(0.1, 30))
_ffi.init(0)
_ffi.set_option("iir_two_pass", -1)
xd = _ffi.DeviceArray(n, np.float32).fill_noise(7)
yd = _ffi.DeviceArray(n, np.float32)
k = _ffi.IirKernel(_ffi.F32, sos=sos)
for _ in range(20): k.filter_dev(xd, yd)
_ffi.sync()
os.environ["SKDSP_FUSED_TRACE"] = out
k.filter_dev(xd, yd); _ffi.sync()
t = np.loadtxt(out, delimiter=",", dtype=np.uint64)
st = t[:, :14].astype(np.float64) * 0.01  # 100 MHz -> us
t0 = st[:, 0].min()
names = ["A", "S", "L", "C", "B0c", "B0s", "B1c", "B1s", "B2c", "B2s", "B3c", "B3s", "end"]
d = np.diff(st, axis=1)
print("segments", len(t), " span %.1f us" % (st[:, 13].max() - t0))
print("phase durations (us): median / p10 / p90")
for i, nm in enumerate(names):
    print("  %-4s %6.2f %6.2f %6.2f" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("  segment total: %.2f" % np.median(st[:, 13] - st[:, 0]))
print("  shader clocks per us over a segment (s_memtime / s_memrealtime): median %.0f  p10 %.0f  p90 %.0f" % tuple(np.percentile(t[:, 14].astype(np.float64) / (st[:, 13] - st[:, 0]), [50, 10, 90])))
wg = (t[:, 15] >> np.uint64(32)).astype(int)
for w in (0, 1, 300):
    rows = np.where(wg == w)[0]
    print("wg", w, "segments", rows.tolist(), "starts", np.round(st[rows, 0] - t0, 1).tolist())
