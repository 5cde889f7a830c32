#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REAL reference.

Runs ONLY in the dev container, where the reference is importable from
/root/reference/src (it is Python, so it cannot travel to the GPU box in any
form).  Output: small .npz / .json fixtures next to this script; they are data
(inputs + the reference's outputs), never reference source.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_golden.py

Fixture groups follow SURVEY.md section 8(c):
  G1  sigsys.upsample            (sigsys.py:3031-3053)
  G2  sigsys.downsample          (sigsys.py:3056-3083) + error conventions
  G3  sigsys.cic                 (sigsys.py:62-93)
  G4  multirate_FIR.filter, 127 taps          (multirate_helper.py:104-109)
  G5  multirate_FIR.filter, 1024 taps, 2^15 c64 (crosses several OLS tiles)
  G6  multirate_FIR.up/dn, 512 taps           (multirate_helper.py:112-127)
  G7  multirate_IIR.filter/up/dn, elliptic SOS (multirate_helper.py:159-192)
  G8  rate_change.up/dn                       (multirate_helper.py:54-83)
  G9  KAT restatements: interp24/deci24 (tests/test_sigsys.py:617-653),
      os_filter 20-sample vector (tests/test_sigsys.py:688-696)
  G10 dtype matrix (which input dtype -> which output dtype)
"""
import json
import os
import sys
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")

import numpy as np  # noqa: E402

from sk_dsp_comm import multirate_helper as mrh  # noqa: E402
from sk_dsp_comm import sigsys as ss  # noqa: E402
from sk_dsp_comm import fir_design_helper as fir_d  # noqa: E402
from sk_dsp_comm import iir_design_helper as iir_d  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(2026)


def cplx(n, dtype=np.complex64):
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).astype(dtype)


def real(n, dtype=np.float32):
    return rng.standard_normal(n).astype(dtype)


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    sz = os.path.getsize(os.path.join(HERE, name))
    print("%-28s %8.1f KB" % (name, sz / 1024.0))


# ---------------------------------------------------------------- G1 upsample
g1 = {}
for dt in ("int64", "float32", "float64", "complex64"):
    for n in (1, 5, 64, 1000):
        x = (np.arange(1, n + 1) * (1 + (0.5j if dt == "complex64" else 0))).astype(dt)
        for L in (1, 2, 3, 4, 12, 2.9):
            y = ss.upsample(x, L)
            key = "%s_n%d_L%s" % (dt, n, str(L).replace(".", "p"))
            g1["x_" + key] = x
            g1["y_" + key] = y
save("g1_upsample.npz", **g1)

# -------------------------------------------------------------- G2 downsample
g2 = {}
g2_err = []
for n in (0, 2, 11, 1000):
    x = real(n, np.float64)
    xc = cplx(n, np.complex64)
    g2["x_n%d" % n] = x
    g2["xc_n%d" % n] = xc
    for M in (1, 2, 3, 12):
        for p in list(range(M)) + [-1]:
            key = "n%d_M%d_p%s" % (n, M, str(p).replace("-", "m"))
            try:
                y = np.ascontiguousarray(ss.downsample(x, M, p))
                yc = np.ascontiguousarray(ss.downsample(xc, M, p))
                g2["y_" + key] = y
                g2["yc_" + key] = yc
            except Exception as e:  # record the convention
                g2_err.append({"n": n, "M": M, "p": p, "type": type(e).__name__, "msg": str(e)})
for bad_M, xlen in ((3.0, 0), (np.int64(3), 9), (2.5, 10)):
    try:
        ss.downsample(np.zeros(xlen), bad_M)
    except Exception as e:
        g2_err.append({"n": xlen, "M": repr(bad_M), "p": 0, "type": type(e).__name__, "msg": str(e)})
try:
    ss.downsample(np.zeros(6), 3, 3)
except Exception as e:
    g2_err.append({"n": 6, "M": 3, "p": 3, "type": type(e).__name__, "msg": str(e)})
try:
    ss.downsample(np.zeros(6), 0)
except Exception as e:
    g2_err.append({"n": 6, "M": 0, "p": 0, "type": type(e).__name__, "msg": str(e)})
save("g2_downsample.npz", **g2)

# --------------------------------------------------------------------- G3 cic
g3 = {}
for (m, k) in ((10, 1), (10, 2), (4, 7), (4, 1), (1, 1), (64, 5), (8, 0), (3, 3)):
    g3["cic_%d_%d" % (m, k)] = ss.cic(m, k)
save("g3_cic.npz", **g3)

# -------------------------------------------------- G4 127-tap FIR .filter
b127 = fir_d.firwin_lpf(127, 0.1)
f127 = mrh.multirate_FIR(b127)
x4r = real(8192)
x4c = cplx(8192)
save("g4_fir127.npz", b=b127, xr=x4r, yr=f127.filter(x4r), xc=x4c, yc=f127.filter(x4c),
     N_forder=np.int64(f127.N_forder))

# -------------------------------------------------- G5 1024-tap FIR .filter
b1024 = fir_d.firwin_lpf(1024, 0.1)
f1024 = mrh.multirate_FIR(b1024)
x5 = cplx(2 ** 15)
x5r = real(20000)
# complex taps as well (frequency-shifted lowpass): exercises the c64 x c64 product
b1024c = b1024 * np.exp(2j * np.pi * 0.11 * np.arange(1024))
f1024c = mrh.multirate_FIR(b1024c)
save("g5_fir1024.npz", b=b1024, x=x5, y=f1024.filter(x5), xr=x5r, yr=f1024.filter(x5r),
     bc=b1024c, yc=f1024c.filter(x5[:12000]))

# ------------------------------------------------------ G6 512-tap up / dn
b512 = fir_d.firwin_lpf(512, 0.1125)
f512 = mrh.multirate_FIR(b512)
x6 = cplx(4096)
x6r = real(3001)
save("g6_fir512_updn.npz", b=b512, x=x6,
     up4=f512.up(x6, 4), dn3=np.ascontiguousarray(f512.dn(x6, 3)),
     up4_dn3=np.ascontiguousarray(ss.downsample(f512.up(x6, 4), 3)),
     up_default=f512.up(x6[:700]), dn_default=np.ascontiguousarray(f512.dn(x6)),
     xr=x6r, upr5=f512.up(x6r, 5), dnr7=np.ascontiguousarray(f512.dn(x6r, 7)))

# -------------------------------------------------------- G7 elliptic SOS IIR
_, _, sos8 = iir_d.IIR_bpf(0.19, 0.2, 0.3, 0.31, 0.5, 60, 1.0, 'ellip', status=False)
_, _, sos7 = iir_d.IIR_bpf(23000, 24000, 28000, 29000, 0.5, 70, 96000, 'ellip', status=False)
i8 = mrh.multirate_IIR(sos8)
i7 = mrh.multirate_IIR(sos7)
x7 = real(2 ** 14)
x7c = cplx(5000)
save("g7_iir_sos.npz", sos8=sos8, sos7=sos7, x=x7,
     y8=i8.filter(x7), up2=i8.up(x7[:6000], 2), dn3=np.ascontiguousarray(i8.dn(x7, 3)),
     y7=i7.filter(x7), xc=x7c, y8c=i8.filter(x7c),
     N_forder8=np.float64(i8.N_forder), N_forder7=np.float64(i7.N_forder))

# -------------------------------------------------------------- G8 rate_change
g8 = {}
x8 = real(2048)
x8c = cplx(1500)
g8["x"] = x8
g8["xc"] = x8c
for tag, args in (("m4", (4,)), ("m12", (12,)), ("m4_cheby", (4, 0.8, 6, 'cheby1'))):
    rc = mrh.rate_change(*args)
    g8[tag + "_b"] = rc.b
    g8[tag + "_a"] = rc.a
    g8[tag + "_up"] = rc.up(x8)
    g8[tag + "_dn"] = np.ascontiguousarray(rc.dn(x8))
    g8[tag + "_upc"] = rc.up(x8c)
    g8[tag + "_dnc"] = np.ascontiguousarray(rc.dn(x8c))
save("g8_rate_change.npz", **g8)

# ------------------------------------------------------------------- G9 KATs
m2 = ss.m_seq(2)
m3 = ss.m_seq(3)
y24 = ss.interp24(m2)
y24b = ss.interp24(m3)
d24 = ss.deci24(y24b)
n = np.arange(0, 20)
xos = np.cos(2 * np.pi * 0.05 * n)
yos = ss.os_filter(xos, np.ones(10), 2 ** 10)
# os_filter / oa_filter on a longer complex input (the reference keeps only the real part)
xosc = cplx(3000, np.complex128)
hos = fir_d.firwin_lpf(65, 0.15)
save("g9_kat.npz", m2=m2, m3=m3, interp24_m2=y24, interp24_m3=y24b, deci24=np.ascontiguousarray(d24),
     os_x=xos, os_b=np.ones(10), os_y=yos, oa_y=ss.oa_filter(xos, np.ones(10), 2 ** 10),
     osc_x=xosc, osc_h=hos, osc_os=ss.os_filter(xosc, hos, 256), osc_oa=ss.oa_filter(xosc, hos, 256))

# ------------------------------------------------------------ G10 dtype matrix
dtm = {}
for dt in ("float32", "float64", "complex64", "complex128", "int32"):
    x = np.ones(24, dtype=dt)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dtm[dt] = {
            "upsample": str(ss.upsample(x, 2).dtype),
            "downsample": str(ss.downsample(x, 2).dtype),
            "fir_filter": str(f127.filter(x).dtype),
            "fir_up": str(f127.up(x, 2).dtype),
            "fir_dn": str(f127.dn(x, 2).dtype),
            "iir_filter": str(i8.filter(x).dtype),
            "iir_up": str(i8.up(x, 2).dtype),
            "iir_dn": str(i8.dn(x, 2).dtype),
            "rc_up": str(mrh.rate_change(4).up(x).dtype),
            "rc_dn": str(mrh.rate_change(4).dn(x).dtype),
        }
# f32 sos + f32 x stays f32 in the reference (scipy result_type)
dtm["float32_sos32"] = {"iir_filter": str(mrh.multirate_IIR(sos8.astype(np.float32)).filter(np.ones(8, np.float32)).dtype)}

misc = {"dtype_matrix": dtm, "downsample_errors": g2_err, "errors": {}}
for name, fn in (("fir_filter_empty", lambda: f127.filter(np.zeros(0))),
                 ("iir_filter_empty", lambda: i8.filter(np.zeros(0))),
                 ("upsample_list", lambda: ss.upsample([1, 2, 3], 2)),
                 ("upsample_2d", lambda: ss.upsample(np.zeros((2, 3)), 2)),
                 ("upsample_L0", lambda: ss.upsample(np.zeros(4), 0)),
                 ("iir_bad_sos", lambda: mrh.multirate_IIR(np.ones((2, 5))).filter(np.zeros(4))),
                 ("rate_change_bad_ftype", lambda: mrh.rate_change(4, ftype='bessel').up(np.zeros(4)))):
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fn()
        misc["errors"][name] = None
    except Exception as e:
        misc["errors"][name] = {"type": type(e).__name__, "msg": str(e)}
# 2-D input: lfilter works along the last axis
x2d = real(3 * 50).reshape(3, 50)
np.savez_compressed(os.path.join(HERE, "g10_2d.npz"), x=x2d, y_fir=f127.filter(x2d), y_iir=i8.filter(x2d))
with open(os.path.join(HERE, "g10_conventions.json"), "w") as f:
    json.dump(misc, f, indent=1, sort_keys=True)
print("g10_conventions.json written")
print(json.dumps(misc["errors"], indent=1))
