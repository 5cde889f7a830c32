"""Pins oracle/ (the CPU restatement) against golden vectors captured from the
REAL reference (tests/golden/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from conftest import GOLDEN, rel_err


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_g1_upsample_exact():
    g = load("g1_upsample.npz")
    keys = [k[2:] for k in g.files if k.startswith("x_")]
    assert len(keys) >= 90
    for k in keys:
        L = float(k.split("_L")[1].replace("p", "."))
        L = int(L) if L == int(L) else L
        y = orc.upsample(g["x_" + k], L)
        ref = g["y_" + k]
        assert y.dtype == ref.dtype, k
        assert np.array_equal(y, ref), k


def test_g2_downsample_exact_and_errors():
    g = load("g2_downsample.npz")
    keys = [k[2:] for k in g.files if k.startswith("y_")]
    assert len(keys) >= 80
    for k in keys:
        n, M, p = k.split("_")
        n, M, p = int(n[1:]), int(M[1:]), int(p[1:].replace("m", "-"))
        y = orc.downsample(g["x_n%d" % n], M, p)
        assert y.dtype == g["y_" + k].dtype and np.array_equal(y, g["y_" + k]), k
        yc = orc.downsample(g["xc_n%d" % n], M, p)
        assert yc.dtype == g["yc_" + k].dtype and np.array_equal(yc, g["yc_" + k]), k
    conv = json.load(open(os.path.join(GOLDEN, "g10_conventions.json")))
    for e in conv["downsample_errors"]:
        if e["type"] == "TypeError":
            with pytest.raises(TypeError, match="M must be an int"):
                orc.downsample(np.zeros(e["n"]), eval(e["M"], {"np": np}))
    with pytest.raises(IndexError):
        orc.downsample(np.zeros(6), 3, 3)
    with pytest.raises(ZeroDivisionError):
        orc.downsample(np.zeros(6), 0)


def test_g3_cic_exact():
    g = load("g3_cic.npz")
    for k in g.files:
        _, m, kk = k.split("_")
        b = orc.cic(int(m), int(kk))
        assert np.array_equal(b, g[k]), k
    # reference KATs (tests/test_sigsys.py:13-26)
    assert np.sum(np.ones(10) / 10 - orc.cic(10, 1)) == 0


def test_g4_fir127():
    g = load("g4_fir127.npz")
    for xk, yk in (("xr", "yr"), ("xc", "yc")):
        y = orc.fir_filter(g["b"], g[xk])
        assert y.dtype == g[yk].dtype
        assert rel_err(y, g[yk])[0] < 1e-12


def test_g5_fir1024():
    g = load("g5_fir1024.npz")
    assert rel_err(orc.fir_filter(g["b"], g["x"]), g["y"])[0] < 1e-12
    assert rel_err(orc.fir_filter(g["b"], g["xr"]), g["yr"])[0] < 1e-12
    assert rel_err(orc.fir_filter(g["bc"], g["x"][:12000]), g["yc"])[0] < 1e-12


def test_g6_fir512_updn():
    g = load("g6_fir512_updn.npz")
    b, x = g["b"], g["x"]
    up4 = orc.fir_up(b, x, 4)
    assert up4.dtype == g["up4"].dtype and rel_err(up4, g["up4"])[0] < 1e-12
    assert rel_err(orc.fir_dn(b, x, 3), g["dn3"])[0] < 1e-12
    assert rel_err(orc.downsample(up4, 3), g["up4_dn3"])[0] < 1e-12
    assert rel_err(orc.fir_up(b, x[:700]), g["up_default"])[0] < 1e-12
    assert rel_err(orc.fir_dn(b, x), g["dn_default"])[0] < 1e-12
    assert rel_err(orc.fir_up(b, g["xr"], 5), g["upr5"])[0] < 1e-12
    assert rel_err(orc.fir_dn(b, g["xr"], 7), g["dnr7"])[0] < 1e-12


def test_g7_iir_sos_bit_exact():
    g = load("g7_iir_sos.npz")
    x = g["x"]
    assert np.array_equal(orc.sos_filter(g["sos8"], x), g["y8"])
    assert np.array_equal(orc.sos_filter(g["sos7"], x), g["y7"])
    assert np.array_equal(orc.sos_up(g["sos8"], x[:6000], 2), g["up2"])
    assert np.array_equal(orc.sos_dn(g["sos8"], x, 3), g["dn3"])
    assert np.array_equal(orc.sos_filter(g["sos8"], g["xc"]), g["y8c"])
    # python loop == C loop
    assert np.array_equal(orc.sos_filter_py(g["sos8"], x[:300]), g["y8"][:300])


def test_g8_rate_change_bit_exact():
    g = load("g8_rate_change.npz")
    for tag, M in (("m4", 4), ("m12", 12), ("m4_cheby", 4)):
        b, a = g[tag + "_b"], g[tag + "_a"]
        for suffix, x in (("", g["x"]), ("c", g["xc"])):
            up = orc.rate_change_up(b, a, M, x)
            dn = orc.rate_change_dn(b, a, M, x)
            assert np.array_equal(up, g[tag + "_up" + suffix]), (tag, suffix)
            assert np.array_equal(dn, g[tag + "_dn" + suffix]), (tag, suffix)


def test_g9_kats():
    """Reference KATs restated: tests/test_sigsys.py:617-653 (interp24/deci24), :688-696 (os_filter)."""
    sig = pytest.importorskip("scipy.signal")
    g = load("g9_kat.npz")

    def interp24(x):
        y = x
        for L in (2, 3, 4):
            b, a = sig.butter(10, 1.0 / L)
            y = orc.lfilter(b, a, L * orc.upsample(y, L))
        return y

    def deci24(x):
        y = x
        for M in (2, 3, 4):
            b, a = sig.butter(10, 1.0 / M)
            y = orc.downsample(orc.lfilter(b, a, y), M)
        return y

    y = interp24(g["m2"])
    np.testing.assert_almost_equal(y, g["interp24_m2"])
    assert rel_err(y, g["interp24_m2"])[0] < 1e-12
    # the hard-coded reference values (first/last of the 72-value KAT)
    np.testing.assert_almost_equal(y[[0, 1, 70, 71]], [8.95202944e-11, 1.34163933e-09, 8.29598667e-01, 8.69628354e-01])
    yd = deci24(interp24(g["m3"]))
    np.testing.assert_almost_equal(yd, [3.33911797e-22, 3.71880014e-10, 4.33029514e-06, 1.16169513e-03,
                                        4.34891180e-02, 4.08255952e-01, 1.16839852e+00])
    assert rel_err(yd, g["deci24"])[0] < 1e-10
    # os_filter KAT == plain FIR of ones(10) over a cosine
    np.testing.assert_almost_equal(orc.fir_filter(g["os_b"], g["os_x"]), g["os_y"])
    np.testing.assert_almost_equal(orc.fir_filter(g["os_b"], g["os_x"]), g["oa_y"])
    # os_filter / oa_filter keep only the real part of a complex result (sigsys.py:534,592)
    ref = np.real(orc.fir_filter(g["osc_h"], g["osc_x"]))
    assert rel_err(ref, g["osc_os"])[0] < 1e-12 and rel_err(ref, g["osc_oa"])[0] < 1e-12


def test_oracle_vs_scipy_when_available():
    sig = pytest.importorskip("scipy.signal")
    rng = np.random.default_rng(7)
    x = rng.standard_normal(3000) + 1j * rng.standard_normal(3000)
    b = rng.standard_normal(33)
    assert rel_err(orc.fir_filter(b, x), sig.lfilter(b, [1], x))[0] < 1e-13
    sos = sig.iirdesign(0.2, 0.3, 1, 60, ftype="ellip", output="sos")
    assert np.array_equal(orc.sos_filter(sos, x.real), sig.sosfilt(sos, x.real))
    bb, aa = sig.butter(7, 0.3)
    assert np.array_equal(orc.lfilter(bb, aa, x), sig.lfilter(bb, aa, x))
    # hist semantics (used by the sharding tests): filtering the tail with history == slice of the whole
    y = orc.fir_filter(b, x)
    y2 = orc.fir_filter(b, x[1000:], hist=x[1000 - 32:1000])
    assert rel_err(y2, y[1000:])[0] < 1e-13


def test_g10_2d_and_dtypes():
    g = load("g10_2d.npz")
    b = load("g4_fir127.npz")["b"]
    assert rel_err(orc.fir_filter(b, g["x"]), g["y_fir"])[0] < 1e-12
    assert np.array_equal(orc.sos_filter(load("g7_iir_sos.npz")["sos8"], g["x"]), g["y_iir"])
