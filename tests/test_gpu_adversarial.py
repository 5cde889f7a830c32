"""Coherent-input parity tests of the float32 engines (run with -m gpu on an MI355X).

Every other float32 / complex64 check drives the kernels with Gaussian noise, where rounding errors
average out.  The two engines with re-associated float32 arithmetic -- the 8192-point float32 FFT of the
overlap-save tile (csrc/ols_core.hpp) and the 3-way bf16 split of the matrix-pipe direct form
(csrc/fir_bx.hip) -- are driven here with inputs whose errors add up instead: DC, a pass-band tone, and a
full-scale tone next to a -100 dB tone, through all-positive taps (sigsys.cic(64, 5), a 1024-tap boxcar,
a 4097-tap boxcar) and the 1024-tap window-design lowpass; through .filter, .up(., 12), .dn(., 12) and the
fused 4/3 resampler; each through the overlap-save engine (complex64 and the two-real-tiles float32 variant)
and the direct engines.  Checker: the float64 oracle.  Bound: 1e-6 on max-abs error / max-abs reference AND
relative L2 (BASELINE.json north_star); reference path sigsys.py:62-93 feeding multirate_helper.py:104-127.
"""
import numpy as np
import pytest

from sk_dsp_comm_amd import _ffi, multirate_helper as mrh, sigsys as ss
from oracle import oracle as orc
from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL32 = 1e-6


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    _ffi.init()
    assert "gfx950" in _ffi.device_info()["name"]
    yield


def firwin_lowpass(ntaps, cutoff):
    m = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = cutoff * np.sinc(cutoff * m) * np.hamming(ntaps)
    return h / np.sum(h)


TAPS = {
    "cic64x5": lambda: ss.cic(64, 5),                  # 316 all-positive taps, unit DC gain
    "box1024": lambda: np.ones(1024) / 1024,
    "firwin1024": lambda: firwin_lowpass(1024, 0.2),
    "box4097": lambda: np.ones(4097) / 4097,           # the longest filter the overlap-save tile takes
}


def signal_of(kind, n, cplx):
    k = np.arange(n, dtype=np.float64)
    f0, f1 = 0.0123, 0.0391                             # both inside every pass band used here
    if kind == "dc":
        x = np.ones(n) * ((1 + 1j) / np.sqrt(2) if cplx else 1.0)
    elif kind == "tone":
        x = np.exp(2j * np.pi * f0 * k) if cplx else np.cos(2 * np.pi * f0 * k)
    elif kind == "tone_m100dB":
        x = (np.exp(2j * np.pi * f0 * k) + 1e-5 * np.exp(2j * np.pi * f1 * k + 0.3j)) if cplx else \
            (np.cos(2 * np.pi * f0 * k) + 1e-5 * np.cos(2 * np.pi * f1 * k + 0.3))
    else:
        raise ValueError(kind)
    return x.astype(np.complex64 if cplx else np.float32)


def check(y, ref, what):
    e_max, e_l2 = rel_err(y, ref)
    assert e_max <= TOL32 and e_l2 <= TOL32, "%s: max/peak %.3g, rel-L2 %.3g > 1e-6" % (what, e_max, e_l2)
    return e_max, e_l2


ENGINES = [("ols", _ffi.FIR_OLS), ("direct", _ffi.FIR_DIRECT)]


@pytest.mark.parametrize("kind", ["dc", "tone", "tone_m100dB"])
@pytest.mark.parametrize("taps", sorted(TAPS))
@pytest.mark.parametrize("cplx", [True, False], ids=["c64", "f32"])
@pytest.mark.parametrize("engine,algo", ENGINES)
def test_filter_coherent_inputs(engine, algo, cplx, taps, kind):
    """.filter: overlap-save (complex64 tile / two real tiles per complex tile) and the direct engines
    (bf16x3 matrix pipe up to 12 lag blocks, FP32 matrix pipe / sliding-window kernels beyond)."""
    b = TAPS[taps]()
    n = 3 * 8192 + 1234 if len(b) > 2000 else 6 * 8192 + 777
    x = signal_of(kind, n, cplx)
    k = _ffi.FirKernel(b, _ffi.code_of(x.dtype))
    k.set_algo(algo)
    y = k.filter(x)
    check(y, orc.fir_filter(b, x), "%s %s %s %s" % (engine, x.dtype.name, taps, kind))


@pytest.mark.parametrize("kind", ["dc", "tone", "tone_m100dB"])
@pytest.mark.parametrize("taps", ["cic64x5", "box1024", "firwin1024"])
@pytest.mark.parametrize("cplx", [True, False], ids=["c64", "f32"])
def test_up_dn_updn_coherent_inputs(cplx, taps, kind):
    """.up(., 12), .dn(., 12) (both engines) and the fused 4/3 resampler on coherent inputs."""
    b = TAPS[taps]()
    n = 4 * 8192 + 12 * 5
    x = signal_of(kind, n, cplx)
    code = _ffi.code_of(x.dtype)
    tag = "%s %s %s" % (x.dtype.name, taps, kind)
    k = _ffi.FirKernel(b, code)
    xs = x[:6000]
    check(k.up(xs, 12), orc.fir_up(b, xs, 12), "up12 " + tag)
    ref_dn = orc.fir_dn(b, x, 12)
    for engine, algo in ENGINES:
        kk = _ffi.FirKernel(b, code)
        kk.set_algo(algo)
        check(kk.dn(x, 12), ref_dn, "dn12/%s %s" % (engine, tag))
    xs = x[:12000]
    check(k.updn(xs, 4, 3), orc.downsample(orc.fir_up(b, xs, 4), 3), "updn43 " + tag)


@pytest.mark.parametrize("m,kk", [(64, 5), (10, 2), (4, 7), (128, 3)])
@pytest.mark.parametrize("kind", ["dc", "tone"])
def test_multirate_fir_of_cic_taps_end_to_end(m, kk, kind):
    """multirate_FIR(cic(m, k)) through the reference surface (a-3 feeding a-5 .. a-7): float32 and complex64
    inputs, reference dtypes out."""
    b = ss.cic(m, kk)
    f = mrh.multirate_FIR(b)
    for cplx in (False, True):
        x = signal_of(kind, 50_000, cplx)
        tag = "cic(%d,%d) %s %s" % (m, kk, kind, x.dtype.name)
        y = f.filter(x)
        assert y.dtype == (np.complex128 if cplx else np.float64)
        check(y, orc.fir_filter(b, x), "filter " + tag)
        check(f.dn(x, m), orc.fir_dn(b, x, m), "dn " + tag)
        check(f.up(x[:4000], m), orc.fir_up(b, x[:4000], m), "up " + tag)


@pytest.mark.parametrize("kind", ["dc", "tone_m100dB"])
def test_full_size_dc_and_tone_windows(kind):
    """Headline size (2^26 complex64, 1024 taps, overlap-save): a coherent input over the whole vector, windows
    against the oracle (steady state: every tile sees the same coherent error pattern)."""
    n = 1 << 26
    b = firwin_lowpass(1024, 0.2)
    k = _ffi.FirKernel(b, _ffi.C64)
    xd = _ffi.DeviceArray(n, np.complex64)
    w = 1 << 20
    for s0 in range(0, n, w):                              # written in pieces: no 512 MiB host temporary
        seg = signal_of(kind, w, True) if kind == "dc" else None
        if seg is None:
            kk = np.arange(s0, s0 + w, dtype=np.float64)
            seg = (np.exp(2j * np.pi * 0.0123 * kk) + 1e-5 * np.exp(2j * np.pi * 0.0391 * kk + 0.3j)).astype(np.complex64)
        xd.write(seg, at=s0)
    yd = _ffi.DeviceArray(n, np.complex64)
    k.filter_dev(xd, yd)
    _ffi.sync()
    for s0 in (0, 7168 * 4000 - 500, n - 20000):
        cnt = 16000
        lo = max(s0 - 1023, 0)
        ref = orc.fir_filter(b, xd.to_host(lo, s0 - lo + cnt))[s0 - lo:]
        check(yd.to_host(s0, cnt), ref, "%s window @%d" % (kind, s0))
    xd.free()
    yd.free()
