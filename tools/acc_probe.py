"""Developer probe: is the f32 accumulation of the matrix-pipe FIR biased?  All-positive random taps and samples (every partial sum grows
monotonically), forced direct form, error against float64: mean (bias) and spread of the relative error vs the number of taps.
Run on the GPU box: python tools/acc_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
from sk_dsp_comm_amd import _ffi
_ffi.init(0)
rng = np.random.default_rng(1)
for M in (1, 12):
    for N in (64, 128, 256, 512, 1024, 1400):
        b = rng.uniform(0.5, 1.0, N) / N
        x = rng.uniform(0.5, 1.0, 120000).astype(np.float32)
        k = _ffi.FirKernel(b, _ffi.code_of(np.float32)); k.set_algo(_ffi.FIR_DIRECT)
        y = k.filter(x) if M == 1 else k.dn(x, M)
        ref = np.convolve(x.astype(np.float64), b)[:len(x)][::M][:len(y)]
        lo = 4000 // M
        e = (y[lo:].astype(np.float64) - ref[lo:]) / ref[lo:]
        print("M=%2d N=%5d: rel err mean %+.2e  rms %.2e  max|.| %.2e" % (M, N, e.mean(), np.sqrt((e**2).mean()), np.abs(e).max()), flush=True)
