#!/usr/bin/env python3
"""G14: golden vectors for the round-2 wideners, captured by running the REAL reference in the dev container (data only).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_golden_wideners.py

  sigsys.os_filter / oa_filter (mode=1)  (sigsys.py:482-598)   filtered signal + per-frame diagnostic matrix
  sigsys.env_det, am_tx, am_rx           (sigsys.py:2784-2942) AM case study around interp24 / deci24
  digitalcom.qam_bb, mpsk_bb, rz_bits, gmsk_bb, time_delay (constant delay)   (digitalcom.py:418-492, 585-667, 998-1048, 1089-1131)
Random transmitters are captured under a fixed np.random.seed: the mirror draws its symbols with the same calls in the
same order.
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
rng = np.random.default_rng(1414)
out = {}

x = rng.standard_normal(300)
h = rng.standard_normal(9) / 3
for name, fn in (("os", ss.os_filter), ("oa", ss.oa_filter)):
    y, ym = fn(x, h, 32, mode=1)
    out[name + "_x"], out[name + "_h"], out[name + "_y"], out[name + "_ymat"] = x, h, y, ym

m = np.cos(2 * np.pi * 1000 / 8000. * np.arange(400))
x192, t192, m24 = ss.am_tx(m, 0.8, fc=75e3)
out["am_m"], out["am_x192"], out["am_t192"], out["am_m24"] = m, x192, t192, m24
m_rx8, t8, m_rx192, x_edet = ss.am_rx(x192)
out["am_rx8"], out["am_t8"], out["am_rx192"], out["am_edet"] = m_rx8, t8, m_rx192, x_edet

np.random.seed(2024)
xq, bq, dq = dc.qam_bb(300, 8, '16qam', 'src', 0.25)
out["qam_x"], out["qam_b"], out["qam_d"] = xq, bq, dq
np.random.seed(2025)
xq, bq, dq = dc.qam_bb(200, 4, 'qpsk', 'rect')
out["qpsk_x"], out["qpsk_b"], out["qpsk_d"] = xq, bq, dq
np.random.seed(2026)
xm, bm, dm = dc.mpsk_bb(256, 10, 8, 'rc', 0.35, 5)
out["mpsk_x"], out["mpsk_b"], out["mpsk_d"] = xm, bm, dm
np.random.seed(2027)
xm, bm, dm = dc.mpsk_bb(128, 6, 4, 'rect')
out["mpsk4_x"], out["mpsk4_b"], out["mpsk4_d"] = xm, bm, dm
np.random.seed(2028)
xr, br, dr = dc.rz_bits(500, 12, 'src', 0.5, 4)
out["rz_x"], out["rz_b"], out["rz_d"] = xr, br, dr
np.random.seed(2029)
yg, dg = dc.gmsk_bb(400, 8, 1, 0.3)
out["gmsk_y"], out["gmsk_d"] = yg, dg
np.random.seed(2030)
yg, dg = dc.gmsk_bb(300, 6, 0)
out["msk_y"], out["msk_d"] = yg, dg
xt = rng.standard_normal(2000)
out["td_x"] = xt
out["td_y"] = dc.time_delay(xt, 1.37, 4)
out["td_y2"] = dc.time_delay(xt, 2.0, 6)
# the time-varying branch (digitalcom.py:1132-1160): a delay per sample, swept and jittered across [1, n-2]
td_d = 2.0 + 0.95 * np.sin(np.arange(2000) * 0.013) ** 2 + 0.04 * rng.random(2000)
td_d[::11] = 1.0
out["td_d"] = td_d
out["td_y3"] = dc.time_delay(xt, td_d, 4)
td_d6 = 1.0 + 3.99 * rng.random(2000)
out["td_d6"] = td_d6
out["td_y4"] = dc.time_delay(xt, td_d6, 6)

np.savez_compressed(os.path.join(HERE, "g14_wideners.npz"), **out)
print("wrote g14_wideners.npz:", {k: np.asarray(v).shape for k, v in out.items()})
