#!/usr/bin/env python3
"""G12: coefficient-header text produced by the REAL reference (coeff2header.py:42-155), kept as
data (input arrays + the exact file text) for the byte-for-byte writer test and the loader test.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_golden_headers.py
"""
import json
import os
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")

import numpy as np  # noqa: E402
import scipy.signal as signal  # noqa: E402

from sk_dsp_comm import coeff2header as c2h  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def text_of(fn, *args):
    with tempfile.NamedTemporaryFile("r", suffix=".h") as f:
        fn(f.name, *args)
        return open(f.name).read()


cases = []
n = np.arange(0, 501)
kat = 3 * np.cos(2 * np.pi * 1000 / 48000 * n) + 2 * np.sin(2 * np.pi * 400 / 48000 * n)  # the reference's own KAT input
for name, h in (("kat501", kat), ("len1", np.array([0.5])), ("len3", np.array([0.25, -0.5, 0.125])),
                ("len4", np.array([1.0, -2.0, 3.5, 1e-13])), ("firwin127", signal.firwin(127, 0.2)),
                ("len9", signal.firwin(9, 0.3))):
    cases.append({"kind": "fir", "name": name, "h": [float(v) for v in h], "text": text_of(c2h.fir_header, h)})
for name, h in (("firwin64", signal.firwin(64, 0.25)), ("len8", signal.firwin(8, 0.4)), ("len17", signal.firwin(17, 0.1)),
                ("len1", np.array([0.999]))):
    cases.append({"kind": "fix", "name": name, "h": [float(v) for v in h], "text": text_of(c2h.fir_fix_header, h)})
for name, sos in (("ellip8", signal.ellip(8, 0.5, 60, [0.2, 0.4], btype="bandpass", output="sos")),
                  ("butter2", signal.butter(2, 0.3, output="sos")), ("cheby5", signal.cheby1(5, 1, 0.45, output="sos"))):
    cases.append({"kind": "sos", "name": name, "sos": [[float(v) for v in r] for r in sos], "text": text_of(c2h.iir_sos_header, sos)})
json.dump(cases, open(os.path.join(HERE, "g12_headers.json"), "w"))
print(len(cases), "cases,", os.path.getsize(os.path.join(HERE, "g12_headers.json")) // 1024, "KiB")
