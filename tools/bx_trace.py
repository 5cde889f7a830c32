"""Developer aid: phase timeline of fir_bx_kernel on BASELINE config 3 (library built with -DSK_BX_TRACE_BUILD):
   tools/build_variant.sh bxtr fir_bx.hip -DSK_BX_TRACE_BUILD
   SKDSP_LIB=.../libskdsp_hip_bxtr.so python tools/bx_trace.py [out.bin]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/bx_trace.bin"
from sk_dsp_comm_amd import _ffi
n = 1 << 26
_ffi.init(0)
k = _ffi.FirKernel(bench.firwin_lowpass(512, 0.225), _ffi.C64)
xd = _ffi.DeviceArray(n, np.complex64).fill_noise(1)
yd = _ffi.DeviceArray(n * 4 // 3, np.complex64)
for _ in range(30): k.updn_dev(xd, yd, 4, 3)
_ffi.sync()
os.environ["SKDSP_BX_TRACE"] = out
k.updn_dev(xd, yd, 4, 3); _ffi.sync()
del os.environ["SKDSP_BX_TRACE"]
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 64, 4, 10)
nwg = t.shape[0]
st = t[..., :8].astype(np.float64)
valid = t[..., 7] != 0
its = valid[:, :, 0].sum(axis=1)
print("workgroups", nwg, " iterations per workgroup: min %d max %d" % (its.min(), its.max()))
names = ["tiles but last (MFMA + stores)", "last tile MFMA", "barrier 1", "split window w+1", "stores of last tile", "barrier 2", "load issue w+2"]
d = np.diff(st, axis=3)
sel = valid & (np.arange(64)[None, :, None] >= 3) & (np.arange(64)[None, :, None] < 38)
print("phase durations (shader clocks): median / p10 / p90")
for i, nm in enumerate(names):
    v = d[..., i][sel]
    print("  %-32s %7.0f %7.0f %7.0f" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
lw = (t[..., 8].astype(np.float64) - st[..., 3])[sel & (t[..., 8] != 0)]
print("  of the split: wait for the window  %7.0f %7.0f %7.0f" % (np.median(lw), np.percentile(lw, 10), np.percentile(lw, 90)))
tot = (st[:, 1:, :, 0] - st[:, :-1, :, 0])[sel[:, 1:, :] & sel[:, :-1, :]]
print("  iteration (top to top)           %7.0f %7.0f %7.0f" % (np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
gap = (st[:, 1:, :, 0] - st[:, :-1, :, 7])[sel[:, 1:, :] & sel[:, :-1, :]]
print("  loop back edge                   %7.0f" % np.median(gap))
# phase relation of the two workgroups of a CU: offset of the loop tops modulo the iteration time
hw = t[:, 5, 0, 9]
cu = ((hw >> np.uint64(32)) & np.uint64(15)).astype(int) * 1024 + ((hw >> np.uint64(13)) & np.uint64(7)).astype(int) * 64 + ((hw >> np.uint64(12)) & np.uint64(1)).astype(int) * 16 + ((hw >> np.uint64(8)) & np.uint64(15)).astype(int)
print("distinct CUs seen:", len(set(cu.tolist())))
per = np.median(tot)
offs = []
for c in set(cu.tolist()):
    w = np.where(cu == c)[0]
    if len(w) == 2:
        for itn in (10, 20, 30):
            a, b = st[w[0], itn, 0, 0], st[w[1], itn, 0, 0]
            # MFMA window of each: [top, stamp 2]; overlap fraction of the two MFMA phases
            offs.append(((b - a) % per) / per)
offs = np.array(offs)
print("phase offset of the two workgroups of a CU (fraction of an iteration): histogram over 10 bins")
print("  ", np.histogram(offs, bins=10, range=(0, 1))[0].tolist())
for itn in (0, 1, 2, 3, 5, 10, 20, 40):
    dd = st[256:512, itn, 0, 0] - st[0:256, itn, 0, 0]
    print("  iteration %2d: workgroup b+256 behind workgroup b by %7.0f clocks (median)" % (itn, np.median(dd)))
# one CU in detail
c = cu[0]; w = np.where(cu == c)[0]
print("CU of workgroup 0 holds workgroups", w.tolist())
t0 = st[w[0], 10, 0, 0]
for g in w:
    for itn in (10, 11):
        print("  wg %4d it %d wave 0 stamps:" % (g, itn), np.round(st[g, itn, 0, :] - t0).astype(int).tolist())
        print("  wg %4d it %d wave 3 stamps:" % (g, itn), np.round(st[g, itn, 3, :] - t0).astype(int).tolist())
