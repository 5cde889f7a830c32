#!/usr/bin/env python3
"""G11: golden vectors for the callers around the hot path, captured by running
the REAL reference in the dev container (same rules as gen_golden.py: data only, never source).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_golden_callers.py

  sigsys.interp24 / deci24            (sigsys.py:2945-3028)  multistage (b,a) interpolation/decimation
  sigsys.ten_band_eq_filt / peaking   (sigsys.py:96-141, 202-260)  ten cascaded peaking biquads
  sigsys.rc_imp / sqrt_rc_imp         (sigsys.py:1847-1945)  pulse shapes (incl. their singular points)
  sigsys.nrz_bits2                    (sigsys.py:2163-2211)  lfilter(b, 1, zero-stuffed +-1 data)
  digitalcom.qam_gray_encode_bb       (digitalcom.py:1584-1681)  lfilter(b, 1, upsample(x_IQ, ns))
  digitalcom.mpsk_gray_encode_bb      (digitalcom.py:1742-1826)
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")

import numpy as np  # noqa: E402

from sk_dsp_comm import sigsys as ss  # noqa: E402
from sk_dsp_comm import digitalcom as dc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(1111)
out = {}

# --- interp24 / deci24 on vectors long enough to cross many scan chunks
xi = rng.standard_normal(1000)
out["i24_x"] = xi
out["i24_y"] = ss.interp24(xi)
xic = (rng.standard_normal(400) + 1j * rng.standard_normal(400))
out["i24c_x"] = xic
out["i24c_y"] = ss.interp24(xic)
xd = rng.standard_normal(24000)
out["d24_x"] = xd
out["d24_y"] = np.ascontiguousarray(ss.deci24(xd))

# --- ten-band equaliser
gdb = np.array([6.0, -3.0, 0.0, 4.5, -8.0, 2.0, 10.0, -12.0, 1.5, 3.0])
xe = rng.standard_normal(8000)
out["eq_gdb"] = gdb
out["eq_x"] = xe
out["eq_y"] = ss.ten_band_eq_filt(xe, gdb)
out["eq_y_q2"] = ss.ten_band_eq_filt(xe, gdb, Q=2.0)
pk = [ss.peaking(g, f, q) for g, f, q in ((5.0, 500.0, 3.5), (-5.0, 500.0, 4.0), (12.0, 16000.0, 3.5))]
out["peak_b"] = np.array([p[0] for p in pk])
out["peak_a"] = np.array([p[1] for p in pk])

# --- pulse shapes (alpha = 0.5 / 0.25 with these Ns put samples exactly on the singular points)
for tag, ns, al, m in (("a", 10, 0.35, 6), ("b", 8, 0.5, 4), ("c", 4, 0.25, 6), ("d", 16, 0.25, 3)):
    out["rc_" + tag] = ss.rc_imp(ns, al, m)
    out["src_" + tag] = ss.sqrt_rc_imp(ns, al, m)
out["pulse_params"] = np.array([[10, 0.35, 6], [8, 0.5, 4], [4, 0.25, 6], [16, 0.25, 3]])

# --- NRZ with user data
bits = rng.integers(0, 2, 300)
out["nrz_bits"] = bits
for pulse in ("rect", "rc", "src"):
    x, b = ss.nrz_bits2(bits, 10, pulse, 0.25, 6)
    out["nrz_x_" + pulse] = x
    out["nrz_b_" + pulse] = b

# --- Gray-coded QAM / MPSK transmitters with external data
data = rng.integers(0, 2, 960)
out["tx_data"] = data
for mod in (2, 4, 16, 64, 256):
    for pulse, ns in (("src", 8), ("rect", 4)):
        x, b, d = dc.qam_gray_encode_bb(None, ns, mod, pulse, 0.35, 6, data)
        out["qam%d_%s_x" % (mod, pulse)] = x
        out["qam%d_%s_b" % (mod, pulse)] = b
x1, b1, d1 = dc.qam_gray_encode_bb(None, 1, 16, "rect", 0.35, 6, data)
out["qam16_ns1_x"] = x1
for mod in (2, 4, 8, 16, 32):
    x, b, d = dc.mpsk_gray_encode_bb(None, 8, mod, "src", 0.25, 6, data)
    out["mpsk%d_x" % mod] = x
    out["mpsk%d_b" % mod] = b
xr, br, dr = dc.mpsk_gray_encode_bb(None, 5, 8, "rc", 0.35, 4, data)
out["mpsk8_rc_x"] = xr

np.savez_compressed(os.path.join(HERE, "g11_callers.npz"), **out)
print("g11_callers.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "g11_callers.npz")) // 1024, "KiB")
