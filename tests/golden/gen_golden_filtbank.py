#!/usr/bin/env python3
"""G13: sigsys.fft_filt_bank (sigsys.py:2588-2694) outputs from the REAL reference (data only).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_golden_filtbank.py
"""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")

import numpy as np  # noqa: E402
import scipy.signal as signal  # noqa: E402

from sk_dsp_comm import sigsys as ss  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(1313)
out = {}
h = signal.firwin(65, 0.2)
xr = rng.standard_normal(3000)                       # 3000 = 23*128 + 56: the tail stays zero
xc = rng.standard_normal(2100) + 1j * rng.standard_normal(2100)
hc = h * np.exp(1j * 0.05 * np.arange(65))            # complex taps
out.update(h=h, xr=xr, xc=xc, hc=hc)
with contextlib.redirect_stdout(io.StringIO()) as buf:
    y, f, fd = ss.fft_filt_bank(xr, h + 0j, n_fft2=128, n_bands2=2, bs=200, fs=1000)
    out.update(odd_y=y, odd_f=f, odd_fd=fd)
    y, f, fd = ss.fft_filt_bank(xr, h + 0j, n_fft2=128, n_bands2=2, bs=200, fs=1000, n_band_odd=False)
    out.update(even_y=y, even_f=f, even_fd=fd)
    y, f, fd = ss.fft_filt_bank(xc, hc, n_fft2=100, n_bands2=1, bs=130, fs=1000)   # non power-of-two FFT
    out.update(cplx_y=y, cplx_f=f, cplx_fd=fd)
out["stdout"] = np.array(buf.getvalue())
np.savez_compressed(os.path.join(HERE, "g13_filtbank.npz"), **out)
print(os.path.getsize(os.path.join(HERE, "g13_filtbank.npz")) // 1024, "KiB")
