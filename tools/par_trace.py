"""Developer aid: phase timeline of iir_par_kernel on BASELINE config 4 (library built with -DSK_PAR_TRACE_BUILD):
   tools/build_variant.sh partr iir_par.hip -DSK_PAR_TRACE_BUILD
   SKDSP_LIB=.../libskdsp_hip_partr.so python tools/par_trace.py [out.bin]
Stamps per wave segment: 0 entry, 1 ticket + table, 2 loads issued, 3 image + G x done, 4 scan done, 5 look-back done, 6 correction done,
7-10 the four pieces of the recurrence, 11 stores issued.  The two waves that share a SIMD are found through HW_ID."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/par_trace.bin"
from sk_dsp_comm_amd import _ffi
n = 1 << 26
sos = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]
_ffi.init(0)
xd = _ffi.DeviceArray(n, np.float32).fill_noise(7)
yd = _ffi.DeviceArray(n, np.float32)
k = _ffi.IirKernel(_ffi.F32, sos=sos)
for _ in range(30): k.filter_dev(xd, yd)
_ffi.sync()
os.environ["SKDSP_PAR_TRACE"] = out
k.filter_dev(xd, yd); _ffi.sync()
del os.environ["SKDSP_PAR_TRACE"]
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 16)
st = t[:, :12].astype(np.float64)
names = ["ticket + table (barrier)", "issue loads", "image + G x (MFMA)", "scan", "look-back", "correction", "recurrence 1", "recurrence 2 (+ stores 1)",
         "recurrence 3 (+ stores 2)", "recurrence 4 (+ stores 3)", "stores 4"]
d = np.diff(st, axis=1)
ok = (t[:, 11] != 0) & (np.arange(len(t)) > 0)
print("segments", len(t))
print("phase durations (shader clocks): median / p10 / p90")
for i, nm in enumerate(names):
    v = d[ok, i]
    print("  %-28s %7.0f %7.0f %7.0f" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
life = st[ok, 11] - st[ok, 0]
print("  %-28s %7.0f %7.0f %7.0f" % ("wave lifetime", np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
# co-residence: waves on the same (XCC, SE, SH, CU, SIMD); for each wave, how much of its recurrence [6, 10] and of its matrix phase [2, 3]
# overlaps a co-resident wave's recurrence
hw = t[:, 12].astype(np.int64); xcc = t[:, 13].astype(np.int64) & 15
key = (xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 12) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
order = np.argsort(key, kind="stable")
rec_ov, mm_ov, rec_len = [], [], []
both = 0
groups = 0
for kk in np.unique(key[ok]):
    idx = np.where((key == kk) & ok)[0]
    if len(idx) < 2:
        continue
    groups += 1
    iv = sorted((st[i, 6], st[i, 10], st[i, 2], st[i, 3], st[i, 0], st[i, 11]) for i in idx)
    for a_i, (r0, r1, m0, m1, e0, e1) in enumerate(iv):
        ov = mo = 0.0
        for b_i, (q0, q1, _, _, _, _) in enumerate(iv):
            if a_i == b_i:
                continue
            ov += max(0.0, min(r1, q1) - max(r0, q0))
            mo += max(0.0, min(m1, q1) - max(m0, q0))
        rec_ov.append(ov / (r1 - r0)); mm_ov.append(mo / max(m1 - m0, 1.0)); rec_len.append(r1 - r0)
rec_ov = np.array(rec_ov); mm_ov = np.array(mm_ov)
print("SIMDs seen: %d; waves per SIMD over the launch: %.1f" % (groups, ok.sum() / max(groups, 1)))
print("fraction of a wave's recurrence phase during which another wave of its SIMD is also in its recurrence: mean %.2f  median %.2f" % (rec_ov.mean(), np.median(rec_ov)))
print("fraction of a wave's image + G x phase during which another wave of its SIMD is in its recurrence:        mean %.2f  median %.2f" % (mm_ov.mean(), np.median(mm_ov)))
print("recurrence phase length when alone (< 5 %% overlapped): %s   when mostly shared (> 80 %%): %s" % (
    ("%.0f" % np.median(np.array(rec_len)[rec_ov < 0.05])) if (rec_ov < 0.05).any() else "-",
    ("%.0f" % np.median(np.array(rec_len)[rec_ov > 0.8])) if (rec_ov > 0.8).any() else "-"))
# one SIMD in detail
kk = key[ok][0]
idx = [i for i in np.where((key == kk) & ok)[0]]
idx.sort(key=lambda i: st[i, 0])
t0 = st[idx[0], 0]
print("one SIMD, its waves in start order (entry, loads issued, G x done, scan, look-back, correction, recurrence end, stores issued), clocks from the first entry:")
for i in idx[:10]:
    print("   seg %5d:" % i, [int(st[i, j] - t0) for j in (0, 2, 3, 4, 5, 6, 10, 11)])
