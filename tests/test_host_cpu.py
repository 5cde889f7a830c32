"""CPU-only checks of the boundary and the host logic (no compute calls without a GPU):
the C-ABI library loads and exports every symbol include/skdsp.h declares, the product
fails loudly without a device, the Python mirror keeps the reference's signatures and
error conventions, and the OLS index algebra holds (host emulation of the tile)."""
import ctypes
import inspect
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import sk_dsp_comm_amd as sk
from sk_dsp_comm_amd import _ffi, multirate_helper as mrh, sigsys as ss
from conftest import GOLDEN, ROOT


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "skdsp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(skdsp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_ffi.lib_path()), "build libskdsp_hip.so first (__graft_entry__.build())"
    L = ctypes.CDLL(_ffi.lib_path())
    declared = _header_symbols()
    assert len(declared) >= 45
    for name in declared:
        assert hasattr(L, name), "missing export: " + name
    assert sorted(_ffi.SYMBOLS) == declared  # the ctypes layer binds exactly the header
    assert b"gfx950" in ctypes.cast(L.skdsp_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_library_is_gfx950_code_object():
    out = subprocess.run(["strings", "-n", "6", _ffi.lib_path()], stdout=subprocess.PIPE).stdout
    assert b"gfx950" in out
    for kern in (b"ols_tile_kernel", b"fir_poly_kernel", b"iir_chunk_kernel", b"upsample_kernel"):
        assert kern in out


def test_fails_loudly_without_gpu():
    L = _ffi.load()
    if L.skdsp_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.SkdspError, match="no HIP device"):
        mrh.multirate_FIR(np.ones(4) / 4).filter(np.ones(16, np.float32))
    with pytest.raises(_ffi.SkdspError):
        ss.upsample(np.ones(4), 2)


def test_signatures_match_reference():
    assert str(inspect.signature(mrh.rate_change.__init__)) == "(self, M_change=12, fcutoff=0.9, N_filt_order=8, ftype='butter')"
    assert str(inspect.signature(mrh.multirate_FIR.__init__)) == "(self, b)"
    assert str(inspect.signature(mrh.multirate_FIR.up)) == "(self, x, L_change=12)"
    assert str(inspect.signature(mrh.multirate_FIR.dn)) == "(self, x, M_change=12)"
    assert str(inspect.signature(mrh.multirate_IIR.up)) == "(self, x, L_change=12)"
    assert str(inspect.signature(mrh.multirate_IIR.dn)) == "(self, x, M_change=12)"
    assert str(inspect.signature(ss.downsample)) == "(x, M, p=0)"
    assert str(inspect.signature(ss.upsample)) == "(x, L)"
    assert str(inspect.signature(ss.cic)) == "(m, k)"
    assert str(inspect.signature(ss.os_filter)) == "(x, h, N, mode=0)"
    assert str(inspect.signature(ss.oa_filter)) == "(x, h, N, mode=0)"


def test_constructors_and_logging(caplog):
    import logging
    caplog.set_level(logging.INFO)
    b = np.ones(127) / 127
    f = mrh.multirate_FIR(b)
    assert f.N_forder == 127 and f.b is b
    assert "FIR filter taps = 127" in caplog.text
    g = np.load(os.path.join(GOLDEN, "g7_iir_sos.npz"))
    i8 = mrh.multirate_IIR(g["sos8"])
    assert i8.N_forder == float(g["N_forder8"]) and "IIR filter order = 16" in caplog.text
    assert mrh.multirate_IIR(g["sos7"]).N_forder == float(g["N_forder7"])
    pytest.importorskip("scipy")
    g8 = np.load(os.path.join(GOLDEN, "g8_rate_change.npz"))
    rc = mrh.rate_change(4)
    assert rc.M == 4 and rc.fc == 0.45 and rc.N_forder == 8
    np.testing.assert_allclose(rc.b, g8["m4_b"], rtol=1e-12)
    np.testing.assert_allclose(rc.a, g8["m4_a"], rtol=1e-12)
    rc = mrh.rate_change(4, 0.8, 6, 'cheby1')
    np.testing.assert_allclose(rc.a, g8["m4_cheby_a"], rtol=1e-12)
    with pytest.warns(UserWarning, match='ftype must be "butter" or "cheby1"'):
        bad = mrh.rate_change(4, ftype='bessel')
    with pytest.raises(AttributeError, match="has no attribute 'b'"):
        bad.up(np.zeros(4))


def test_error_conventions_before_any_gpu_call():
    conv = json.load(open(os.path.join(GOLDEN, "g10_conventions.json")))
    for e in conv["downsample_errors"]:
        if e["type"] == "TypeError":
            with pytest.raises(TypeError, match="M must be an int"):
                ss.downsample(np.zeros(e["n"]), eval(e["M"], {"np": np}))
    with pytest.raises(IndexError, match="index 3 is out of bounds for axis 1 with size 3"):
        ss.downsample(np.zeros(6), 3, 3)
    with pytest.raises(ZeroDivisionError):
        ss.downsample(np.zeros(6), 0)
    errs = conv["errors"]
    with pytest.raises(AttributeError, match=re.escape(errs["upsample_list"]["msg"])):
        ss.upsample([1, 2, 3], 2)
    with pytest.raises(ValueError, match=re.escape(errs["upsample_2d"]["msg"])):
        ss.upsample(np.zeros((2, 3)), 2)
    with pytest.raises(ValueError, match=re.escape(errs["upsample_L0"]["msg"])):
        ss.upsample(np.zeros(4), 0)
    with pytest.raises(ValueError, match=re.escape(errs["fir_filter_empty"]["msg"])):
        mrh.multirate_FIR(np.ones(3)).filter(np.zeros(0))
    with pytest.raises(ValueError, match=re.escape(errs["iir_bad_sos"]["msg"])):
        mrh.multirate_IIR(np.ones((2, 5))).filter(np.zeros(4))
    sos = np.array([[1.0, 0, 0, 2.0, 0, 0]])
    with pytest.raises(ValueError, match=re.escape("sos[:, 3] should be all ones")):
        mrh.multirate_IIR(sos).filter(np.zeros(4))
    with pytest.raises(TypeError, match="M must be an int"):
        mrh.multirate_FIR(np.ones(3)).dn(np.zeros(9), 3.0)
    # empty / tiny inputs that the reference answers without filtering
    assert ss.downsample(np.zeros(2), 3).shape == (0,)
    assert ss.upsample(np.zeros(0), 3).shape == (0,) and ss.upsample(np.zeros(0), 3).dtype == np.float64


def test_cic_golden_exact():
    g = np.load(os.path.join(GOLDEN, "g3_cic.npz"))
    for k in g.files:
        _, m, kk = k.split("_")
        assert np.array_equal(ss.cic(int(m), int(kk)), g[k]), k
    # reference KATs tests/test_sigsys.py:13-26, tests/test_digitalcom.py:39-62
    assert np.sum(np.ones(10) / 10 - ss.cic(10, 1)) == 0
    correct = [0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.07, 0.08, 0.09, 0.1,
               0.09, 0.08, 0.07, 0.06, 0.05, 0.04, 0.03, 0.02, 0.01]
    assert np.sum(correct - ss.cic(10, 2)) == 0
    assert len(ss.cic(4, 7)) == 22


def test_ols_tile_index_algebra_host_emulation(tmp_path):
    """Compiles csrc/ols_core.hpp for the HOST and runs one whole overlap-save tile."""
    exe = str(tmp_path / "ols_emul")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "scikit-dsp-comm_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "ols_emul.cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE).stdout.decode()
    assert out.strip().endswith("OK"), out


@pytest.mark.parametrize("name", ["ols4k_emul", "ols2k_emul"])
def test_interpolator_decimator_tiles_host_emulation(tmp_path, name):
    """Compiles csrc/ols4k_core.hpp (4096 points, 16 per thread) / csrc/ols2k_core.hpp (2048 points, 8 per thread) for the HOST and
    runs whole tiles of the frequency-domain interpolator (multirate_FIR.up: one forward transform, L products + inverse transforms)
    and decimator (multirate_FIR.dn: M forward transforms accumulated, one inverse) with the tables the library builds, against the
    float64 polyphase sums."""
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "scikit-dsp-comm_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", name + ".cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE).stdout.decode()
    assert out.strip().endswith("OK"), out


def test_tf2sos_factorisation_matches_reference_tf_outputs():
    """skdsp_tf_create runs (b,a) as biquads (host-only factorisation, no GPU needed): the
    factored cascade, evaluated by the oracle's sosfilt, must reproduce the REFERENCE's
    lfilter(b,a,.) outputs (rate_change golden vectors, interp24 KAT) far inside 1e-6."""
    from oracle import oracle as orc
    from conftest import rel_err
    g = np.load(os.path.join(GOLDEN, "g8_rate_change.npz"))
    for tag, M in (("m4", 4), ("m12", 12), ("m4_cheby", 4)):
        sos = _ffi.tf2sos(g[tag + "_b"], g[tag + "_a"])
        assert sos.shape[1] == 6 and (sos[:, 3] == 1).all()
        up = orc.sos_filter(sos, M * orc.upsample(g["x"], M))
        assert rel_err(up, g[tag + "_up"])[0] < 1e-8, tag
        dn = orc.downsample(orc.sos_filter(sos, g["xc"]), M)
        assert rel_err(dn, g[tag + "_dnc"])[0] < 1e-8, tag
    # degenerate shapes: odd order, more zeros than poles, leading-zero numerator (pure delay)
    for b, a in (([0.2, 0.3], [1, -0.5, 0.2, 0.1]), ([0, 0, 1.0, 0.5], [1, -0.9]), ([1.0], [2.0, -1.0]),
                 ([0.5, 0.2, 0.1, 0.7, 0.3], [1.0, 0.1])):
        x = np.random.default_rng(1).standard_normal(500)
        assert rel_err(orc.sos_filter(_ffi.tf2sos(b, a), x), orc.lfilter(b, a, x))[0] < 1e-12
    with pytest.raises(ValueError):
        _ffi.tf2sos([1.0], [0.0, 1.0])


def test_package_surface():
    for name in ("rate_change", "multirate_FIR", "multirate_IIR", "upsample", "downsample", "cic"):
        assert hasattr(sk, name)
    assert sk.config.strict_dtype is True


# ---------------------------------------------------------------------------------------------
# callers around the hot path: the host-side pieces against the G11 fixtures
# ---------------------------------------------------------------------------------------------
def _g11():
    return np.load(os.path.join(ROOT, "tests", "golden", "g11_callers.npz"))


def test_pulse_designs_match_reference():
    from sk_dsp_comm_amd import sigsys
    g = _g11()
    for tag, (ns, al, m) in zip("abcd", g["pulse_params"]):
        ns, m = int(ns), int(m)
        assert np.allclose(sigsys.rc_imp(ns, al, m), g["rc_" + tag], rtol=1e-13, atol=1e-15)
        assert np.allclose(sigsys.sqrt_rc_imp(ns, al, m), g["src_" + tag], rtol=1e-13, atol=1e-15)


def test_peaking_matches_reference():
    from sk_dsp_comm_amd import sigsys
    g = _g11()
    for i, (gd, fc, q) in enumerate(((5.0, 500.0, 3.5), (-5.0, 500.0, 4.0), (12.0, 16000.0, 3.5))):
        b, a = sigsys.peaking(gd, fc, q)
        assert np.allclose(b, g["peak_b"][i], rtol=1e-13) and np.allclose(a, g["peak_a"][i], rtol=1e-13)


def test_gray_symbol_mapping_matches_reference_without_gpu():
    """ns = 1 returns the mapped symbols before any filtering: pins the Gray LUT / bit order."""
    from sk_dsp_comm_amd import digitalcom as dc
    g = _g11()
    x, b, d = dc.qam_gray_encode_bb(None, 1, 16, "rect", 0.35, 6, g["tx_data"])
    assert b == 1 and np.array_equal(d, g["tx_data"][:len(d)])
    assert np.allclose(x, g["qam16_ns1_x"], rtol=0, atol=1e-15)
    assert list(dc._gray_lut(5)) == [0, 1, 3, 2, 7, 6, 4, 5, 15, 14, 12, 13, 8, 9, 11, 10, 31, 30, 28, 29, 24, 25, 27, 26,
                                     16, 17, 19, 18, 23, 22, 20, 21]
    with pytest.raises(ValueError):
        dc.qam_gray_encode_bb(None, 4, 8, ext_data=g["tx_data"])
    with pytest.raises(ValueError):
        dc.mpsk_gray_encode_bb(None, 4, 64, ext_data=g["tx_data"])


# ---------------------------------------------------------------------------------------------
# coefficient header formats against the text the reference itself wrote (G12)
# ---------------------------------------------------------------------------------------------
def test_coefficient_headers_byte_exact_and_round_trip(tmp_path):
    import json
    from sk_dsp_comm_amd import coeff2header as c2h
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "g12_headers.json")))
    assert {c["kind"] for c in cases} == {"fir", "fix", "sos"}
    for c in cases:
        fn = str(tmp_path / (c["kind"] + "_" + c["name"] + ".h"))
        if c["kind"] == "fir":
            h = np.array(c["h"])
            c2h.fir_header(fn, h)
            assert open(fn).read() == c["text"], c["name"]
            assert np.allclose(c2h.read_fir_header(fn), h, rtol=0, atol=0.5e-12)
        elif c["kind"] == "fix":
            h = np.array(c["h"])
            c2h.fir_fix_header(fn, h)
            assert open(fn).read() == c["text"], c["name"]
            assert np.array_equal(c2h.read_fir_header(fn) * 2 ** 15, np.rint(h * 2 ** 15))
        else:
            sos = np.array(c["sos"])
            c2h.iir_sos_header(fn, sos)
            assert open(fn).read() == c["text"], c["name"]
            back = c2h.read_sos_header(fn)
            assert back.shape == sos.shape and np.allclose(back, sos / sos[:, 3:4], rtol=1e-6, atol=1e-12)


# ----------------------------------------------------------------- round-2 additions
def test_kernel_cache_follows_reassigned_coefficients():
    """The reference reads obj.b / obj.sos on every call: reassigning them must rebuild the device handles
    (multirate_helper._KernelCache keys on a fingerprint of the coefficient arrays)."""
    from sk_dsp_comm_amd import multirate_helper as mrh
    made = []
    holder = {"b": np.arange(4.0)}
    cache = mrh._KernelCache(lambda code: made.append((code, holder["b"].copy())) or len(made), lambda: (holder["b"],))
    k1 = cache.get(np.float32)
    assert cache.get(np.float32) == k1 and len(made) == 1
    assert cache.get(np.complex64) != k1 and len(made) == 2
    holder["b"] = np.arange(4.0) + 1j            # real -> complex taps of the same length
    k3 = cache.get(np.float32)
    assert k3 != k1 and len(made) == 3 and np.iscomplexobj(made[-1][1])
    holder["b"][0] = 5.0                         # in-place edit is seen too
    assert cache.get(np.float32) != k3 and len(made) == 4
    f = mrh.multirate_FIR(np.ones(3))
    assert f._bc is False
    f.b = np.ones(3) * 1j
    assert f._bc is True


def test_options_table_without_gpu():
    """skdsp_set_option / skdsp_get_option work without a device; unknown names are BADARG -> ValueError."""
    from sk_dsp_comm_amd import _ffi
    assert _ffi.get_option("ols_reserve") == 8
    # the A/B switches of the round-5 kernels and their defaults: on
    for name, default in (("fir_dn_fold", 1), ("fir_up_rep", 1), ("iir_up_jump", 1), ("iir_up_lean", 1), ("iir_dn_t96", 1), ("iir_seq", 1)):
        assert _ffi.get_option(name) == default, name
    with _ffi.option("iir_planar", 1):
        assert _ffi.get_option("iir_planar") == 1
    assert _ffi.get_option("iir_planar") == 0
    with pytest.raises(ValueError):
        _ffi.set_option("no_such_switch", 1)


def test_bench_self_launch_reaches_the_device_layer():
    """`python bench.py --gpus 2` with no launcher must spawn its own ranks and fail (here: no GPU) inside the
    library's device binding, not in the launcher; the non-zero exit code of a rank is the exit code."""
    import subprocess
    import sys
    from sk_dsp_comm_amd import _ffi
    if _ffi.load().skdsp_device_count() > 0:
        pytest.skip("a GPU is present: the launcher path is covered by the gpu tests")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode()
    assert out.returncode != 0
    assert "no HIP device available" in text and "must be launched" not in text, text[-2000:]


@pytest.mark.parametrize("n,L,M,hist,lg", [(1_000_003, 1, 1, 1023, 16), (999_999, 1, 3, 511, 14), (300_001, 4, 3, 128, 12),
                                              (70_000, 12, 1, 86, 11), (5, 1, 1, 1023, 10), (2 ** 20, 1, 12, 767, 16)])
def test_host_chunk_planner(n, L, M, hist, lg):
    """The chunk planner of the host-pointer pipeline (no GPU needed): chunks tile the input, chunk starts are multiples
    of M (output phase 0 stays aligned), each chunk reaches back `hist` samples (less at the start of the vector), the
    output ranges tile [0, floor(n L / M)) and equal the outputs whose newest input lies in the chunk."""
    from sk_dsp_comm_amd import _ffi
    plan = _ffi.host_chunk_plan(n, L, M, hist, lg)
    assert plan[0][0] == 0 and plan[-1][1] == n and plan[0][3] == 0 and plan[-1][4] == (n * L) // M
    for k, (ib, ie, ih, ob, oe) in enumerate(plan):
        assert ib % M == 0 and ie > ib and ih == min(hist, ib)
        if k:
            assert ib == plan[k - 1][1] and ob == plan[k - 1][4]
        assert ob == (ib * L) // M and (oe == (ie * L) // M or k == len(plan) - 1)
        # output m = needs inputs up to floor(m M / L): inside this chunk for every m of the range
        if oe > ob:
            assert ib <= (ob * M) // L and ((oe - 1) * M) // L < ie


# ---- parallel-form expansion of the IIR cascade (host side of csrc/iir_par.hip; no GPU) -----------------------------
def _par_designs():
    from scipy import signal
    sos8 = np.load(os.path.join(GOLDEN, "g7_iir_sos.npz"))["sos8"]
    return {
        "ellip_bpf8": sos8,                                                        # BASELINE config 4
        "butter8_rate_change12": signal.butter(8, 0.9 / 12, output="sos"),         # rate_change(12) (multirate_helper.py:62)
        "cheby1_8_rate_change12": signal.cheby1(8, 0.05, 0.9 / 12, output="sos"),  # rate_change(12, ftype='cheby1') (:64)
        "butter5_odd": signal.butter(5, 0.2, output="sos"),                        # a first-order section (b2 = a2 = 0)
        "ellip_bandstop": signal.ellip(6, 1, 60, [0.2, 0.3], btype="bandstop", output="sos"),
        "single_biquad": signal.butter(2, 0.3, output="sos"),
    }


@pytest.mark.parametrize("name", ["ellip_bpf8", "butter8_rate_change12", "cheby1_8_rate_change12", "butter5_odd", "ellip_bandstop",
                                  "single_biquad"])
def test_parallel_form_expansion_reproduces_the_cascade(name):
    """H(z) = c0 + sum_k (r0_k + r1_k z^-1) / (1 + a1_k z^-1 + a2_k z^-2) as the library expands it (long double, arithmetic
    modulo each denominator) against scipy.signal.sosfilt on an impulse and on noise: float64 roundoff level."""
    from scipy import signal
    sos = _par_designs()[name]
    info = _ffi.sos_par_info(sos)
    assert info["accepted"], info
    assert info["ir_err"] < 1e-12 and info["kappa"] < 100
    rng = np.random.default_rng(5)
    for x in (np.r_[1.0, np.zeros(4095)], rng.standard_normal(20000)):
        ref = signal.sosfilt(sos, x)
        y = info["c0"] * x
        for a1, a2, r0, r1 in info["sections"]:
            y = y + signal.lfilter([r0, r1], [1.0, a1, a2], x)
        assert np.max(np.abs(y - ref)) <= 2e-13 * max(np.max(np.abs(ref)), 1e-300)


def test_float32_from_rest_states_are_earned_per_filter():
    """V32 (csrc/iir_par.hip): float32 / complex64 signals through 7 - 8 biquads may form the chunks' from-rest end states on the float32 matrix
    instruction -- where the plan's probe (the instruction's fmaf chain emulated bit for bit on DC, the Nyquist alternation, a tone on every
    section's resonance, noise) shows less than 5e-7 of output error.  BASELINE config 4's band-pass is admitted; designs whose branches
    cancel more -- among them one with a SMALLER cancellation factor, which is why the test is a measurement and not a norm -- are refused."""
    from scipy import signal
    sos8 = np.load(os.path.join(GOLDEN, "g7_iir_sos.npz"))["sos8"]
    i4 = _ffi.sos_par_info(sos8)
    assert i4["accepted"] and i4["v32_admitted"] and 1e-7 < i4["v32_err"] < 5e-7 and i4["v32_err_t96"] < 5e-7, i4
    refused = {
        "ellip_bpf_0.1_0.2": signal.ellip(8, 0.5, 60, [0.1, 0.2], btype="bandpass", output="sos")[:8],
        "cheby2_highpass14": signal.cheby2(14, 50, 0.4, btype="highpass", output="sos"),
        "butter_bpf16": signal.butter(8, [0.2, 0.3], btype="bandpass", output="sos"),
    }
    for name, sos in refused.items():
        info = _ffi.sos_par_info(sos)
        assert info["accepted"], name                       # the parallel form itself serves them (float64 states)
        assert not info["v32_admitted"] and info["v32_err"] > 5e-7, (name, info["v32_err"], info["kappa"])
    assert _ffi.sos_par_info(refused["ellip_bpf_0.1_0.2"])["kappa"] < i4["kappa"]      # (less cancellation by the norm, more error measured)


def test_parallel_form_refuses_what_it_cannot_expand():
    from scipy import signal
    one = signal.butter(2, 0.3, output="sos")
    twice = np.vstack([one, one])                       # the same pole pair in two sections: no simple-pole expansion
    assert not _ffi.sos_par_info(twice)["accepted"]
    fir_heavy = np.array([[1.0, 0.5, 0.25, 1.0, 0.0, 0.0], [1.0, 0.2, 0.1, 1.0, -0.5, 0.0]])   # numerator degree > denominator degree
    assert not _ffi.sos_par_info(fir_heavy)["accepted"]
    unstable_ok = np.array([[1.0, 0.0, 0.0, 1.0, -1.0, 0.0]])   # an integrator expands (one real pole); the launcher refuses it by its decay test
    assert _ffi.sos_par_info(unstable_ok)["accepted"]


# ---- bench.py's stdout contract: ONE line the driver can keep (BENCH_r05.json: a 22.9 KB line came back parsed = null) ----------
def _bench_module():
    import importlib
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def _recorded_full_record():
    """The full record of a real run: round 5's line (profiles/r05/bench_fir1024.json: headline + 18 other workloads, 22.9 KB)."""
    txt = open(os.path.join(ROOT, "profiles", "r05", "bench_fir1024.json")).read().strip().splitlines()[-1]
    return json.loads(txt)


def test_bench_final_line_of_a_recorded_run_is_small_and_complete():
    bench = _bench_module()
    rec = _recorded_full_record()
    assert len(json.dumps(rec)) > 20000                       # (the record that was not kept)
    fl = bench.final_line(rec)
    line = json.dumps(fl)
    assert len(line) < bench.FINAL_LINE_MAX_BYTES <= 8000
    back = json.loads(line)
    assert back == fl and "\n" not in line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["value"] == rec["value"] and back["ms_per_step"] == rec["ms_per_step"]      # the contract's numbers are not rounded
    r = back["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - rec["roofline"]["frac"]) < 1e-5 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-5
    assert r["algorithmic_bytes_per_launch"] == 16 * 2 ** 26                                # byte counts stay exact
    assert abs(r["traffic"] - rec["roofline"]["traffic"]) <= 1e-5 * r["traffic"]
    c = back["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c and c["unit"] == "MSamples/s"
    assert back["cpu_baseline_scipy"]["kind"] == "scipy"
    assert back["config"]["workload"].startswith("multirate_FIR.filter: 1024-tap") and "model" not in back["config"]
    assert len(back["rows"]) == 19 and all("frac" in v and "ms" in v for v in back["rows"].values())
    assert "other_configs" not in back and not any(k.startswith("row_") for k in r)


def test_bench_final_line_worst_case_still_fits():
    """Every string at its longest, 64 ranks, a failed config-5 leg, forty rows of errors: the line sheds members rather than grow."""
    bench = _bench_module()
    rec = _recorded_full_record()
    rec["n_gpus"] = 64
    rec["config"]["sharding"] = "s" * 5000
    rec["config"]["logical_device"] = "d" * 5000
    rec["config"]["n_ranks_rccl"] = 64
    rec["roofline"]["kernel"] = "k" * 5000
    rec["cpu_baseline"]["sample"] = "c" * 5000
    rec["per_rank"] = {"step_ms": [0.23456789] * 64, "kernel_ms": [0.23456789] * 64}
    rec["halo_fallback_used"], rec["halo_state"], rec["halo_form"] = [0] * 64, [2] * 64, "overlapped (probation passed)"
    rec["parity_halo_max_err"], rec["parity_interior_max_err"], rec["parity_ok"] = 2.5e-7, 2.4e-7, True
    rec["config5"] = {"what": "w" * 3000, "total_samples": 1 << 30, "samples_per_gpu": 1 << 24, "n_gpus": 64, "ms": 0.07, "kernel_ms_max": 0.06,
                      "value": 1.5e7, "unit": "MSamples/s", "frac_per_gpu": 0.5, "speedup_vs_n1": 54.2, "steps": 50,
                      "n1_reference": {"profile": "p" * 400, "ms": 3.8, "value": 282000.0, "kernel_source_sha256": {"a": "b" * 64}},
                      "parity_halo_max_err": 2.5e-7, "parity_interior_max_err": 2.4e-7, "halo_fallback_used": [0] * 64, "halo_state": [2] * 64,
                      "parity_ok": True, "error": "e" * 3000}
    rec["rows"] = {"row%02d" % i: {"error": "x" * 60} for i in range(40)}
    fl = bench.final_line(rec)
    line = json.dumps(fl)
    assert len(line) <= bench.FINAL_LINE_MAX_BYTES
    back = json.loads(line)
    for k in ("metric", "value", "roofline", "cpu_baseline", "config", "n_ranks_rccl", "halo_fallback_used", "parity_ok"):
        assert k in back, k
    assert back["n_ranks_rccl"] == 64 and back["config5"]["n1_cross_check"]["ms"] == 3.8 and back["config5"]["speedup_vs_n1"] == 54.2


def test_bench_emit_prints_exactly_one_stdout_line(tmp_path, monkeypatch, capsys):
    bench = _bench_module()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec = _recorded_full_record()
    bench.emit(rec)
    cap = capsys.readouterr()
    lines = cap.out.strip().split("\n")
    assert len(lines) == 1 and len(lines[0]) <= bench.FINAL_LINE_MAX_BYTES
    out = json.loads(lines[0])
    assert out["roofline"]["frac"] > 0.5 and out["cpu_baseline"]["value"] > 1.0 and out["detail"] == os.path.join("gpurun_out", "bench_detail_n1.json")
    side = json.load(open(os.path.join(str(tmp_path), out["detail"])))
    assert "other_configs" in side and len(side["other_configs"]) == 18                   # the long record is kept -- beside stdout
    detail = [ln for ln in cap.err.split("\n") if ln.startswith("bench.py detail: ")]
    assert len(detail) == 19 and all(json.loads(ln[len("bench.py detail: "):]) for ln in detail)
    assert not any(ln.lstrip().startswith("{") for ln in cap.err.split("\n"))            # nothing on stderr can be taken for the line


def test_bench_descriptor_1_carries_the_line_and_nothing_else(tmp_path):
    """A library that writes to descriptor 1 through C stdio (librccl announces its path there) is flushed when the process exits -- BEHIND the result line.
    bench.claim_stdout() hands descriptor 1 to stderr before anything is loaded; emit() writes to the saved descriptor."""
    code = (
        "import sys, os, json, ctypes\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "bench.ROOT = %r\n"
        "bench.claim_stdout()\n"
        "libc = ctypes.CDLL(None)\n"
        "libc.printf(b'a library announcing itself through C stdio\\n')\n"       # (stays in the C buffer until exit on a pipe)
        "print('a python print behind the claim')\n"
        "rec = json.loads(open(os.path.join(%r, 'profiles', 'r05', 'bench_fir1024.json')).read().strip().splitlines()[-1])\n"
        "bench.emit(rec)\n"
        "os.write(1, b'a raw write to descriptor 1 behind the line\\n')\n" % (ROOT, str(tmp_path), ROOT))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().split("\n") if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["roofline"]["frac"] > 0.5, lines
    err = r.stderr.decode()
    assert "announcing itself" in err and "python print behind the claim" in err and "raw write to descriptor 1" in err
    # a reader of BOTH streams as one (2>&1) finds the line LAST: emit() flushes what libraries left in C stdio buffers before it prints
    merged = subprocess.run([sys.executable, "-c", code.rsplit("os.write", 1)[0]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert merged.returncode == 0
    mlines = [ln for ln in merged.stdout.decode().split("\n") if ln.strip()]
    assert any("announcing itself" in ln for ln in mlines[:-1]) and json.loads(mlines[-1])["roofline"]["frac"] > 0.5, mlines[-3:]


def test_profile_stamps_hash_the_code_not_the_comments(tmp_path, monkeypatch):
    """bench.source_hashes names the kernel sources a profile was collected with; a comment or white-space edit must not change the name, a code edit must."""
    bench = _bench_module()
    a = bench._code_only("int f(int x) {\n    return x + 1;   // one more\n}\n/* block\n comment */\n")
    b = bench._code_only("int f(int x) { return x + 1; }   // reworded note\n")
    c = bench._code_only("int f(int x) { return x + 2; }\n")
    assert a == b and a != c
    h = bench.source_hashes("fir1024")
    assert set(h) == {"fir_ols.hip", "ols_core.hpp", "careful.hpp"} and all(isinstance(v, str) and len(v) == 16 for v in h.values())


def test_profile_reduction_reports_the_steady_state(tmp_path):
    """tools/reduce_pmc.py: from a per-dispatch kernel trace, the mean / median of the LAST `steps` launches (the ones bench.py times), not the average over the
    clock-settle launches in front of them; and the HBM bytes of a step from the FETCH_SIZE / WRITE_SIZE passes (KiB; the read side doubled on gfx950)."""
    src = tmp_path / "src"
    dst = tmp_path / "dst"
    src.mkdir(); dst.mkdir()
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp"]
    t = 1000
    durs = [350000] * 50 + [300000] * 50 + [235000] * 400          # ns: a ramping clock, then 400 timed steps
    for i, d in enumerate(durs):
        rows.append('KERNEL_DISPATCH,1,1,7,"void skdsp::ols_tile_kernel<false, false, false, false>(skdsp::OlsArgs)",%d,%d,%d' % (i, t, t + d))
        t += d + 2000
    rows.append('KERNEL_DISPATCH,1,1,9,"void skdsp::fill_noise_kernel<float>(float*)",999,1,500')   # (not a step)
    (src / "kernel_trace_fir1024.csv").write_text("\n".join(rows) + "\n")
    (src / "trace_bench_fir1024.json").write_text(json.dumps({"steps": 400, "ms_per_step": 0.2371, "roofline": {"kernel_ms": 0.2352, "frac": 0.5706}}) + "\n")
    (src / "kernel_stats_fir1024.csv").write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n"ols_tile_kernel",500,1,1,1,1,1,1\n')
    head = '"Correlation_Id","Dispatch_Id","Agent_Id","Kernel_Name","Counter_Name","Counter_Value"\n'
    name = '"void skdsp::ols_tile_kernel<false, false, false, false>(skdsp::OlsArgs)"'
    (src / "pmc_fir1024_FETCH_SIZE.csv").write_text(head + "".join('%d,%d,1,%s,"FETCH_SIZE",%f\n' % (i, i, name, 264550.0) for i in range(20)))
    (src / "pmc_fir1024_WRITE_SIZE.csv").write_text(head + "".join('%d,%d,1,%s,"WRITE_SIZE",%f\n' % (i, i, name, 524288.0) for i in range(20)))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "reduce_pmc.py"), str(src), str(dst), "fir1024"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()
    st = json.load(open(dst / "kernel_steady_fir1024.json"))
    assert st["timed_steps"] == 400 and st["steps_in_trace"] == 500
    assert abs(st["steady_mean_us"] - 235.0) < 1e-6 and abs(st["steady_median_us"] - 235.0) < 1e-6
    assert abs(st["all_launches_avg_us"] - (50 * 350 + 50 * 300 + 400 * 235) / 500.0) < 1e-6          # (what `--stats` would have averaged)
    assert abs(st["frac_of_8TBps_from_steady_mean"] - 16 * 2 ** 26 / 235e-6 / 8e12) < 1e-9
    assert st["same_run_bench_line"]["kernel_ms_hip_events"] == 0.2352
    pm = json.load(open(dst / "pmc_fir1024.json"))
    d = pm["derived"]
    assert abs(d["hbm_read_bytes_per_step"] - 264550.0 * 1024 * 2) < 1 and abs(d["hbm_write_bytes_per_step"] - 524288.0 * 1024) < 1
    assert abs(d["traffic_over_algorithmic"] - (264550.0 * 2048 + 524288.0 * 1024) / (16 * 2 ** 26)) < 1e-9
    assert set(pm["source_sha256"]) == {"fir_ols.hip", "ols_core.hpp", "careful.hpp"}
