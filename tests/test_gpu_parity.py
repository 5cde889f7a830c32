"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(libskdsp_hip.so via ctypes); the CPU oracle and the golden vectors captured from the
real reference are the checkers.

Tolerances (BASELINE.json north_star / SURVEY.md 8c):
  * integer up/down-sample indexing: bit-exact
  * float32 / complex64 filtering: <= 1e-6 (max-abs error / max-abs reference) and rel-L2
  * float64 / complex128 kernels:  <= 1e-11
"""
import os

import numpy as np
import pytest

import sk_dsp_comm_amd as sk
from sk_dsp_comm_amd import _ffi, multirate_helper as mrh, sigsys as ss
from oracle import oracle as orc
from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu

TOL32 = 1e-6
TOL64 = 1e-11


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    _ffi.init()
    info = _ffi.device_info()
    assert "gfx950" in info["name"], info
    yield


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def assert_close(y, ref, tol, what=""):
    e_max, e_l2 = rel_err(y, ref)
    assert e_max <= tol and e_l2 <= tol, "%s: max/peak %.3g, rel-L2 %.3g > %.1g" % (what, e_max, e_l2, tol)


def cnoise(rng, n, dt=np.complex64):
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).astype(dt)


# ----------------------------------------------------------------- resamplers
def test_g1_upsample_bit_exact():
    g = load("g1_upsample.npz")
    for k in [k[2:] for k in g.files if k.startswith("x_")]:
        L = float(k.split("_L")[1].replace("p", "."))
        L = int(L) if L == int(L) else L
        y = ss.upsample(g["x_" + k], L)
        assert y.dtype == g["y_" + k].dtype and np.array_equal(y, g["y_" + k]), k


def test_g2_downsample_bit_exact():
    g = load("g2_downsample.npz")
    for k in [k[2:] for k in g.files if k.startswith("y_")]:
        n, M, p = k.split("_")
        n, M, p = int(n[1:]), int(M[1:]), int(p[1:].replace("m", "-"))
        for xk, yk in (("x_n%d" % n, "y_" + k), ("xc_n%d" % n, "yc_" + k)):
            y = ss.downsample(g[xk], M, p)
            assert y.dtype == g[yk].dtype and np.array_equal(y, g[yk]), k


def test_resample_large_and_other_dtypes():
    rng = np.random.default_rng(1)
    x = rng.integers(-2 ** 31, 2 ** 31 - 1, size=1_000_003, dtype=np.int32)
    assert np.array_equal(ss.downsample(x, 7, 3), orc.downsample(x, 7, 3))
    x16 = rng.integers(-30000, 30000, size=10_001).astype(np.int16)
    y = ss.downsample(x16, 4, 1)
    assert y.dtype == np.int16 and np.array_equal(y, x16[: 2500 * 4].reshape(-1, 4)[:, 1])
    xc = cnoise(rng, 300_001, np.complex128)
    assert np.array_equal(ss.upsample(xc, 5), orc.upsample(xc, 5))
    xf = rng.standard_normal(2 ** 20 + 1).astype(np.float32)
    y = ss.upsample(xf, 3)
    assert y.dtype == np.float64 and np.array_equal(y, orc.upsample(xf, 3))
    # round trip: downsample(upsample(x, L), L) == x
    assert np.array_equal(ss.downsample(ss.upsample(xf, 6), 6), xf.astype(np.float64))


@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64, np.complex128])
def test_downsample_device_blocks_phases_alignment(dt):
    """sigsys.downsample on the device (sigsys.py:3078-3083: x[p::M] cut to floor(n / M) outputs), the kernel that keeps its next block in flight (resample.hip,
    round 6): every stride whose elements share a 64-byte line and a few beyond (the strided gather), every phase, lengths that end inside a block, inside a
    16-byte unit and on a block boundary, 16-byte aligned and element-aligned inputs (the latter take the gather), and nothing written behind the last output."""
    import ctypes
    L = _ffi.load()
    rng = np.random.default_rng(11)
    esz = np.dtype(dt).itemsize
    nmax = 200_003
    x = cnoise(rng, nmax + 8, dt) if np.dtype(dt).kind == "c" else rng.standard_normal(nmax + 8).astype(dt)
    xd = _ffi.DeviceArray.from_host(x)
    yd = _ffi.DeviceArray(nmax // 2 + 64, dt)
    for M in (2, 3, 4, 5, 7, 8, 12, 16, 17):
        for n in (nmax, 65536 * M, 65536 * M + 1, 4096 * M + 5, 70_001):
            if n > nmax:
                continue
            for p in sorted({0, 1, M // 2, M - 1}):
                for off in (0, 1):            # (1: the input pointer is element-aligned only)
                    n_out = n // M
                    _ffi.check(L.skdsp_memset(ctypes.c_void_p(yd.ptr), 0x5a, yd.n * esz))
                    _ffi.check(L.skdsp_downsample_dev(ctypes.c_void_p(xd.ptr + off * esz), n, M, p, _ffi.code_of(dt), ctypes.c_void_p(yd.ptr)))
                    got = yd.to_host(0, n_out)
                    ref = x[off:off + n_out * M].reshape(-1, M)[:, p]
                    assert np.array_equal(got, ref), (np.dtype(dt).name, M, n, p, off)
                    guard = yd.to_host(n_out, 64).view(np.uint8)
                    assert np.all(guard == 0x5a), ("wrote past the last output", np.dtype(dt).name, M, n, p, off)
    xd.free(); yd.free()


def test_fir_bx_tile_runs_are_the_same_outputs():
    """fir_bx.hip, T16 (round 6): the float32 plain filter on the matrix pipe stores its 256-output tiles as runs -- a 4 x 4 transpose between a lane's four result
    registers and the four 16-lane groups (v_permlane32_swap / v_permlane16_swap).  Only where results LAND changes: bit-identical to the per-column stores
    (option fir_bx_t16 = 0) for every tap count the form serves, ragged lengths (the last, partial tile keeps the guarded per-column stores), an output offset
    by one element, and nothing written behind the last output."""
    import ctypes
    L = _ffi.load()
    rng = np.random.default_rng(5)
    for ntaps in (17, 64, 127, 160, 192):
        b = rng.standard_normal(ntaps) / np.sqrt(ntaps)
        k = _ffi.FirKernel(b, _ffi.F32)
        k.set_algo(_ffi.FIR_DIRECT)
        for n in (300_000, 262_144, 65_537, 4_099):
            x = rng.standard_normal(n).astype(np.float32)
            xd = _ffi.DeviceArray.from_host(x)
            yd = _ffi.DeviceArray(n + 65, np.float32)
            outs = []
            _ffi.debug_path()   # (cleared)
            for v in (0, 1):
                for off in (0, 1):
                    with _ffi.option("fir_bx_t16", v):
                        _ffi.check(L.skdsp_memset(ctypes.c_void_p(yd.ptr), 0x5a, yd.n * 4))
                        k.filter_dev(xd, yd.window(off, n), n)
                        outs.append(yd.to_host(off, n))
                        assert np.all(yd.to_host(off + n, 64 - off).view(np.uint8) == 0x5a), (ntaps, n, v, off)
            assert "fir_bx" in _ffi.debug_path()
            for o in outs[1:]:
                assert np.array_equal(outs[0], o), (ntaps, n)
            ref = orc.fir_filter(b, x[:5000])
            assert_close(outs[0][:5000], ref, TOL32, "fir_bx float32 %d taps" % ntaps)
            xd.free(); yd.free()


# ------------------------------------------------------------------------ FIR
def test_g4_fir127():
    g = load("g4_fir127.npz")
    f = mrh.multirate_FIR(g["b"])
    for xk, yk in (("xr", "yr"), ("xc", "yc")):
        y = f.filter(g[xk])
        assert y.dtype == g[yk].dtype
        assert_close(y, g[yk], TOL32, "fir127 " + xk)


@pytest.mark.parametrize("algo", [_ffi.FIR_DIRECT, _ffi.FIR_OLS])
def test_g5_fir1024_both_algorithms(algo):
    g = load("g5_fir1024.npz")
    k = _ffi.FirKernel(g["b"], _ffi.C64)
    k.set_algo(algo)
    assert k.algo_for(len(g["x"])) == algo
    assert_close(k.filter(g["x"]), g["y"], TOL32, "fir1024 c64 algo %d" % algo)
    kc = _ffi.FirKernel(g["bc"], _ffi.C64)
    kc.set_algo(algo)
    assert_close(kc.filter(g["x"][:12000].copy()), g["yc"], TOL32, "fir1024 complex taps algo %d" % algo)


def test_g5_fir1024_mirror_and_real():
    g = load("g5_fir1024.npz")
    f = mrh.multirate_FIR(g["b"])
    y = f.filter(g["x"])
    assert y.dtype == np.complex128
    assert_close(y, g["y"], TOL32, "fir1024 auto")
    assert_close(f.filter(g["xr"]), g["yr"], TOL32, "fir1024 f32 real")
    # float64 callers get float64 kernels
    assert_close(f.filter(g["xr"].astype(np.float64)), g["yr"], TOL64, "fir1024 f64")
    assert_close(f.filter(g["x"].astype(np.complex128)), g["y"], TOL64, "fir1024 c128")


def test_g6_fir512_up_dn_updn():
    g = load("g6_fir512_updn.npz")
    f = mrh.multirate_FIR(g["b"])
    x = g["x"]
    up4 = f.up(x, 4)
    assert up4.dtype == g["up4"].dtype and up4.shape == g["up4"].shape
    assert_close(up4, g["up4"], TOL32, "up4")
    dn3 = f.dn(x, 3)
    assert dn3.shape == g["dn3"].shape
    assert_close(dn3, g["dn3"], TOL32, "dn3")
    assert_close(f.updn(x, 4, 3), g["up4_dn3"], TOL32, "up4/dn3 fused")
    assert_close(ss.downsample(up4, 3), g["up4_dn3"], TOL32, "downsample(up4,3)")
    assert_close(f.up(x[:700]), g["up_default"], TOL32, "up default L=12")
    assert_close(f.dn(x), g["dn_default"], TOL32, "dn default M=12")
    assert_close(f.up(g["xr"], 5), g["upr5"], TOL32, "up5 real")
    assert_close(f.dn(g["xr"], 7), g["dnr7"], TOL32, "dn7 real")


@pytest.mark.parametrize("n,P", [(1, 1), (5, 3), (100, 127), (4097, 48), (7168, 1024), (7169, 1024), (8192, 1025),
                                 (20000, 512), (3 * 7168 + 5, 700), (9000, 2000), (33333, 4097)])
def test_fir_ols_edges_vs_oracle(n, P):
    """tile boundaries, ragged tails, tap counts around the overlap quantum (512)"""
    rng = np.random.default_rng(n * 31 + P)
    x = cnoise(rng, n)
    b = rng.standard_normal(P) / np.sqrt(P)
    ref = orc.fir_filter(b, x)
    k = _ffi.FirKernel(b, _ffi.C64)
    if P >= 2:
        k.set_algo(_ffi.FIR_OLS)
        assert_close(k.filter(x), ref, TOL32, "ols n=%d P=%d" % (n, P))
    k.set_algo(_ffi.FIR_DIRECT)
    assert_close(k.filter(x), ref, TOL32, "direct n=%d P=%d" % (n, P))


@pytest.mark.parametrize("n,P", [(4096, 48), (7168, 127), (7169, 127), (2 * 7680 + 3, 127), (3 * 7168 + 1, 1024),
                                 (50001, 513), (100000, 1025), (1 << 20, 2000)])
def test_fir_ols_real_pairs_vs_oracle(n, P):
    """float32 signals: two real tiles per complex tile (odd tile counts, ragged tails)"""
    rng = np.random.default_rng(n + P)
    x = rng.standard_normal(n).astype(np.float32)
    b = rng.standard_normal(P) / np.sqrt(P)
    ref = orc.fir_filter(b, x)
    k = _ffi.FirKernel(b, _ffi.F32)
    k.set_algo(_ffi.FIR_OLS)
    assert k.algo_for(n) == _ffi.FIR_OLS
    assert_close(k.filter(x), ref, TOL32, "ols real n=%d P=%d" % (n, P))
    # history for the sharded path
    if n > 3 * P:
        cut = n // 3 + 1
        import ctypes
        xd = _ffi.DeviceArray.from_host(x[cut:], headroom=P - 1)
        hist = np.ascontiguousarray(x[cut - (P - 1):cut])
        _ffi.check(_ffi.load().skdsp_memcpy_h2d(ctypes.c_void_p(xd.ptr - hist.nbytes), ctypes.c_void_p(hist.ctypes.data), hist.nbytes))
        yd = _ffi.DeviceArray(n - cut, np.float32)
        k.filter_dev(xd, yd, n_hist=P - 1)
        assert_close(yd.to_host(), ref[cut:], TOL32, "ols real history")


@pytest.mark.parametrize("L,M", [(1, 1), (2, 1), (1, 2), (4, 3), (3, 4), (12, 1), (1, 12), (6, 4), (5, 5), (7, 24)])
def test_fir_polyphase_all_ratios_vs_oracle(L, M):
    rng = np.random.default_rng(L * 100 + M)
    n = 5003
    b = rng.standard_normal(193) / 14
    for x in (cnoise(rng, n), rng.standard_normal(n).astype(np.float32)):
        k = _ffi.FirKernel(b, _ffi.code_of(x.dtype))
        ref = orc.downsample(orc.fir_up(b, x, L), M)
        y = k.updn(x, L, M)
        assert y.shape == ref.shape
        assert_close(y, ref, TOL32, "updn L=%d M=%d %s" % (L, M, x.dtype))


def test_fir_history_matches_slice_of_whole():
    """n_hist semantics of the *_dev entry points (what the sharded path relies on)."""
    rng = np.random.default_rng(5)
    n, P, cut = 60000, 1024, 23456
    x = cnoise(rng, n)
    b = rng.standard_normal(P) / 32
    ref = orc.fir_filter(b, x)
    for algo in (_ffi.FIR_OLS, _ffi.FIR_DIRECT):
        k = _ffi.FirKernel(b, _ffi.C64)
        k.set_algo(algo)
        xd = _ffi.DeviceArray.from_host(x[cut:], headroom=P - 1)
        import ctypes
        hist = np.ascontiguousarray(x[cut - (P - 1):cut])
        _ffi.check(_ffi.load().skdsp_memcpy_h2d(ctypes.c_void_p(xd.ptr - hist.nbytes), ctypes.c_void_p(hist.ctypes.data), hist.nbytes))
        yd = _ffi.DeviceArray(n - cut, np.complex64)
        k.filter_dev(xd, yd, n_hist=P - 1)
        assert_close(yd.to_host(), ref[cut:], TOL32, "history algo %d" % algo)


def test_fir_2d_and_dtype_matrix():
    g = load("g10_2d.npz")
    b = load("g4_fir127.npz")["b"]
    y = mrh.multirate_FIR(b).filter(g["x"])
    assert y.shape == g["y_fir"].shape and y.dtype == np.float64
    assert_close(y, g["y_fir"], TOL32, "2-D")
    import json
    dtm = json.load(open(os.path.join(GOLDEN, "g10_conventions.json")))["dtype_matrix"]
    sos = load("g7_iir_sos.npz")["sos8"]
    f, i8, rc = mrh.multirate_FIR(b), mrh.multirate_IIR(sos), mrh.rate_change(4)
    for dt in ("float32", "float64", "complex64", "complex128", "int32"):
        x = np.ones(24, dtype=dt)
        got = {"upsample": ss.upsample(x, 2), "downsample": ss.downsample(x, 2), "fir_filter": f.filter(x),
               "fir_up": f.up(x, 2), "fir_dn": f.dn(x, 2), "iir_filter": i8.filter(x), "iir_up": i8.up(x, 2),
               "iir_dn": i8.dn(x, 2), "rc_up": rc.up(x), "rc_dn": rc.dn(x)}
        for kname, arr in got.items():
            assert str(arr.dtype) == dtm[dt][kname], (dt, kname, arr.dtype)
    y32 = mrh.multirate_IIR(sos.astype(np.float32)).filter(np.ones(8, np.float32))
    assert str(y32.dtype) == dtm["float32_sos32"]["iir_filter"]


def test_native_dtype_switch():
    g = load("g4_fir127.npz")
    sk.config.strict_dtype = False
    try:
        y = mrh.multirate_FIR(g["b"]).filter(g["xc"])
        assert y.dtype == np.complex64
        assert_close(y, g["yc"], TOL32)
    finally:
        sk.config.strict_dtype = True


# ------------------------------------------------------------------------ IIR
def test_g7_iir_sos():
    g = load("g7_iir_sos.npz")
    i8, i7 = mrh.multirate_IIR(g["sos8"]), mrh.multirate_IIR(g["sos7"])
    x = g["x"]
    y = i8.filter(x)
    assert y.dtype == np.float64
    assert_close(y, g["y8"], TOL32, "sos8 filter")
    assert_close(i7.filter(x), g["y7"], TOL32, "sos7 filter")
    assert_close(i8.up(x[:6000], 2), g["up2"], TOL32, "sos8 up2")
    assert_close(i8.dn(x, 3), g["dn3"], TOL32, "sos8 dn3")
    assert_close(i8.filter(g["xc"]), g["y8c"], TOL32, "sos8 complex")
    # float64 input -> float64 I/O kernels: essentially exact
    assert_close(i8.filter(x.astype(np.float64)), g["y8"], 1e-9, "sos8 f64")


def test_g8_rate_change():
    pytest.importorskip("scipy")
    g = load("g8_rate_change.npz")
    for tag, args in (("m4", (4,)), ("m12", (12,)), ("m4_cheby", (4, 0.8, 6, 'cheby1'))):
        rc = mrh.rate_change(*args)
        assert_close(rc.up(g["x"]), g[tag + "_up"], TOL32, tag + " up")
        assert_close(rc.dn(g["x"]), g[tag + "_dn"], TOL32, tag + " dn")
        assert_close(rc.up(g["xc"]), g[tag + "_upc"], TOL32, tag + " up complex")
        assert_close(rc.dn(g["xc"]), g[tag + "_dnc"], TOL32, tag + " dn complex")


def test_g9_kats_interp24_deci24_through_gpu():
    """tests/test_sigsys.py:617-653 restated on the GPU primitives (TF-form Butterworth order 10)."""
    sig = pytest.importorskip("scipy.signal")
    g = load("g9_kat.npz")

    def interp24(x):
        y = np.asarray(x, dtype=np.float64)
        for L in (2, 3, 4):
            b, a = sig.butter(10, 1.0 / L)
            y = _ffi.IirKernel(_ffi.F64, b=b, a=a).up(np.ascontiguousarray(y), L)
        return y

    def deci24(x):
        y = np.asarray(x, dtype=np.float64)
        for M in (2, 3, 4):
            b, a = sig.butter(10, 1.0 / M)
            y = _ffi.IirKernel(_ffi.F64, b=b, a=a).dn(np.ascontiguousarray(y), M)
        return y

    y = interp24(g["m2"])
    np.testing.assert_almost_equal(y, g["interp24_m2"])
    yd = deci24(interp24(g["m3"]))
    np.testing.assert_almost_equal(yd, g["deci24"])
    np.testing.assert_almost_equal(yd, [3.33911797e-22, 3.71880014e-10, 4.33029514e-06, 1.16169513e-03,
                                        4.34891180e-02, 4.08255952e-01, 1.16839852e+00])
    # os_filter KAT (tests/test_sigsys.py:688-696) == FIR of ones(10)
    np.testing.assert_almost_equal(mrh.multirate_FIR(g["os_b"]).filter(g["os_x"]), g["os_y"])


def test_os_oa_filter_next_row():
    """sigsys.os_filter / oa_filter (sigsys.py:482-598) on the OLS engine."""
    g = load("g9_kat.npz")
    for fn, k in ((ss.os_filter, "os_y"), (ss.oa_filter, "oa_y")):
        y = fn(g["os_x"], g["os_b"], 2 ** 10)
        assert y.dtype == np.float64
        np.testing.assert_almost_equal(y, g[k])
    y = ss.os_filter(g["osc_x"], g["osc_h"], 256)
    assert y.dtype == np.float64 and y.shape == g["osc_os"].shape
    assert_close(y, g["osc_os"], TOL64, "os_filter complex in")
    assert_close(ss.oa_filter(g["osc_x"].astype(np.complex64), g["osc_h"], 256), g["osc_oa"], TOL32, "oa_filter c64")
    y1, ymat = ss.os_filter(g["os_x"], g["os_b"], 1024, mode=1)   # (the diagnostic matrix: test_wideners_match_reference)
    assert ymat.shape[1] == len(y1)
    with pytest.raises(ValueError):
        ss.oa_filter(g["os_x"], g["os_b"], 4)
    assert ss.os_filter(np.zeros(0), g["os_b"], 64).shape == (0,)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 8191, 8192 * 3 + 17, 300_001, 2 ** 22 + 5])
def test_iir_scan_lengths_vs_oracle(n):
    """chunk / workgroup / tail boundaries of the scan"""
    g = load("g7_iir_sos.npz")
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    ref = orc.sos_filter(g["sos8"], x)
    y = _ffi.IirKernel(_ffi.F32, sos=g["sos8"]).filter(x)
    assert_close(y, ref, TOL32, "sos8 n=%d" % n)


def test_iir_slow_decay_and_integrator_exact_scan():
    """The scan is exact, not a warm-up approximation: poles at radius 0.99999 and a pure
    integrator (pole on the unit circle) must still match the sequential recurrence."""
    rng = np.random.default_rng(3)
    n = 500_000
    x = rng.standard_normal(n)
    r = 0.99999
    sos = np.array([[1.0, 0.0, 0.0, 1.0, -2 * r * np.cos(0.01), r * r],
                    [1.0, -1.0, 0.0, 1.0, -0.5, 0.0]])
    assert_close(_ffi.IirKernel(_ffi.F64, sos=sos).filter(x), orc.sos_filter(sos, x), 1e-9, "slow decay")
    integ = np.array([[1.0, 0.0, 0.0, 1.0, -1.0, 0.0]])
    xi = rng.standard_normal(n) * 1e-3
    assert_close(_ffi.IirKernel(_ffi.F64, sos=integ).filter(xi), np.cumsum(xi), 1e-9, "integrator")


def test_iir_many_sections_split():
    sig = pytest.importorskip("scipy.signal")
    sos = sig.butter(30, 0.2, output="sos")  # 15 sections -> 12 + 3
    rng = np.random.default_rng(8)
    x = rng.standard_normal(40_000).astype(np.float32)
    assert_close(mrh.multirate_IIR(sos).filter(x), orc.sos_filter(sos, x), TOL32, "15 sections")


# ------------------------------------------------- full-size property tests
def test_fir_full_size_properties():
    """BASELINE config 2 size (2^26 c64, 1024 taps): spot windows against the oracle
    (the FIR is local) + linearity, device resident."""
    n, P = 2 ** 26, 1024
    g = load("g5_fir1024.npz")
    b = g["b"]
    k = _ffi.FirKernel(b, _ffi.C64)
    xd = _ffi.DeviceArray(n, np.complex64).fill_noise(2026)
    yd = _ffi.DeviceArray(n, np.complex64)
    k.filter_dev(xd, yd)
    _ffi.sync()
    assert k.algo_for(n) == _ffi.FIR_OLS
    rng = np.random.default_rng(0)
    starts = [0, 7168 - 20, n - 5000] + [int(s) for s in rng.integers(P, n - 5000, size=5)]
    for s0 in starts:
        w = 4096
        lo = max(0, s0 - (P - 1))
        xs = xd.to_host(lo, s0 + w - lo)
        ref = orc.fir_filter(b, xs)[s0 - lo:]
        assert_close(yd.to_host(s0, w), ref, TOL32, "window @%d" % s0)
    # linearity: filter(2x) == 2 filter(x) exactly in float arithmetic (power-of-two scale)
    x2 = _ffi.DeviceArray.from_host(2 * xd.to_host(0, 2 ** 20))
    y2 = _ffi.DeviceArray(2 ** 20, np.complex64)
    k.filter_dev(x2, y2)
    assert np.array_equal(y2.to_host(0, 2 ** 19), 2 * yd.to_host(0, 2 ** 19))


def test_iir_full_size_properties():
    """BASELINE config 4 size (2^26 f32, 8 biquads): head against the oracle and
    time-invariance (a delayed input gives the delayed output) deep inside the vector."""
    g = load("g7_iir_sos.npz")
    n = 2 ** 26
    k = _ffi.IirKernel(_ffi.F32, sos=g["sos8"])
    xd = _ffi.DeviceArray(n, np.float32).fill_noise(7)
    yd = _ffi.DeviceArray(n, np.float32)
    k.filter_dev(xd, yd)
    _ffi.sync()
    m = 2 ** 21
    xh = xd.to_host(0, m)
    assert_close(yd.to_host(0, m), orc.sos_filter(g["sos8"], xh), TOL32, "head")
    # window far from the start: the filter forgets (|pole|max = 0.9947) after ~8k samples
    s0 = n - 3 * m
    xs = xd.to_host(s0 - 20000, m + 20000)
    ref = orc.sos_filter(g["sos8"], xs)[20000:]
    assert_close(yd.to_host(s0, m), ref, TOL32, "deep window")


def test_updn_full_size_config3():
    """BASELINE config 3: L=4 / M=3 with the 512-tap prototype on 2^26 c64 (89 478 485 outputs)."""
    g = load("g6_fir512_updn.npz")
    b = g["b"]
    n = 2 ** 26
    n_out = (n * 4) // 3
    assert n_out == 89478485
    k = _ffi.FirKernel(b, _ffi.C64)
    xd = _ffi.DeviceArray(n, np.complex64).fill_noise(11)
    yd = _ffi.DeviceArray(n_out, np.complex64)
    k.updn_dev(xd, yd, 4, 3)
    _ffi.sync()
    for s_in in (0, 12345 * 3, n - 9000):
        s_in -= s_in % 3
        lo = max(0, s_in - 201)  # multiple of 3: keeps output phase 0 aligned
        xs = xd.to_host(lo, min(6000, n - lo))
        ref = orc.downsample(orc.fir_up(b, xs, 4), 3)
        m0 = (s_in * 4) // 3
        skip = ((s_in - lo) * 4) // 3
        w = 4000
        assert_close(yd.to_host(m0, w), ref[skip:skip + w], TOL32, "updn window @%d" % s_in)


@pytest.mark.parametrize("dt", [np.complex64, np.float32])
def test_up_dn_full_size_reference_defaults(dt):
    """multirate_FIR(512-tap prototype).up(x) / .dn(x) at the reference's defaults L_change = M_change = 12 (multirate_helper.py:112-127) on
    2^26 samples of the high rate -- the matrix-pipe kernel with its row tiles dealt to wave pairs (.up) / its lags dealt to the four waves (.dn):
    windows at the start, across window boundaries in the middle and at the end against the oracle."""
    g = load("g6_fir512_updn.npz")
    b = g["b"]
    code = _ffi.code_of(dt)
    k = _ffi.FirKernel(b, code)
    n_hi = 2 ** 26
    # .up: n_hi / 12 inputs -> n_in * 12 outputs
    n_in = n_hi // 12
    xd = _ffi.DeviceArray(n_in, dt).fill_noise(5)
    yd = _ffi.DeviceArray(n_in * 12, dt)
    k.up_dev(xd, yd, 12)
    _ffi.sync()
    for s_in in (0, 1_000_003, n_in // 2 + 77, n_in - 3000):
        lo = max(0, s_in - 64)
        xs = xd.to_host(lo, min(3000, n_in - lo))
        ref = orc.fir_up(b, xs, 12)
        skip = (s_in - lo) * 12
        w = min(20000, len(ref) - skip)
        assert_close(yd.to_host(s_in * 12, w), ref[skip:skip + w], TOL32, "up12 window @%d" % s_in)
    xd.free(); yd.free()
    # .dn: n_hi inputs -> n_hi / 12 outputs
    xd = _ffi.DeviceArray(n_hi, dt).fill_noise(6)
    yd = _ffi.DeviceArray(n_hi // 12, dt)
    k.dn_dev(xd, yd, 12)
    _ffi.sync()
    for s_out in (0, 777_777, n_hi // 24 + 5, n_hi // 12 - 2500):
        s_in = s_out * 12
        lo = max(0, s_in - 12 * 50)   # (a multiple of 12: keeps the kept phase aligned)
        xs = xd.to_host(lo, min(12 * 2600, n_hi - lo))
        ref = orc.fir_dn(b, xs, 12)
        skip = (s_in - lo) // 12
        w = min(2000, len(ref) - skip, n_hi // 12 - s_out)
        assert_close(yd.to_host(s_out, w), ref[skip:skip + w], TOL32, "dn12 window @%d" % s_out)
    xd.free(); yd.free()


# ------------------------------------------------------------- RCCL plumbing on one GPU
def test_rccl_single_rank_communicator_p2p_and_allreduce():
    """The 8-GPU halo path cannot run on a 1-GPU box, but everything except the peer hop can:
    dlopen(librccl), unique id, ncclCommInitRank (1 rank), a grouped send+recv to self on the
    library stream with the exact halo size of config 5 (1023 complex64 = 8184 B, 8-byte
    aligned only), and the 1-element all-reduces bench.py uses for barrier / max."""
    import ctypes
    import subprocess
    import sys
    code = r"""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.join(%r, 'scikit-dsp-comm_amd'))
sys.path.insert(0, %r)
os.environ['SKDSP_DIST_FORCE_COMM'] = '1'  # read once, when the library first needs its options
from sk_dsp_comm_amd import _ffi
_ffi.init(0)
L = _ffi.load()
buf = ctypes.create_string_buffer(128)
_ffi.check(L.skdsp_dist_unique_id(buf))
_ffi.check(L.skdsp_dist_init(0, 1, ctypes.c_char_p(buf.raw)))
rng = np.random.default_rng(0)
x = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
xd = _ffi.DeviceArray.from_host(x, headroom=1023)
n_halo = 1023
src = xd.ptr + (x.size - n_halo) * 8
dst = xd.ptr - n_halo * 8
_ffi.check(L.skdsp_dist_sendrecv(ctypes.c_void_p(src), 0, ctypes.c_void_p(dst), 0, n_halo * 8))
_ffi.sync()
got = np.empty(n_halo, np.complex64)
_ffi.check(L.skdsp_memcpy_d2h(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(dst), got.nbytes))
assert np.array_equal(got, x[-n_halo:]), 'self send/recv mismatch'
d = ctypes.c_double(3.5)
_ffi.check(L.skdsp_dist_allreduce_max(ctypes.byref(d))); assert d.value == 3.5
_ffi.check(L.skdsp_dist_allreduce_sum(ctypes.byref(d))); assert d.value == 3.5
_ffi.check(L.skdsp_dist_barrier())
# the IIR state exchange: ncclAllGather of 17 doubles through the transport's staging buffers
from sk_dsp_comm_amd import sharding
tr = sharding.RcclTransport.__new__(sharding.RcclTransport)
tr.rank, tr.world, tr._rdzv = 0, 1, None
v = rng.standard_normal(17)
tab = tr.allgather_state(v)
assert tab.shape == (1, 17) and np.array_equal(tab[0], v), 'all-gather mismatch'
from scipy import signal
sos = signal.ellip(8, 0.5, 60, [0.2, 0.4], btype='bandpass', output='sos')
xr = rng.standard_normal(100000).astype(np.float32)
iir = sharding.ShardedIIR(sos, tr, dtype=np.float32)
xd2 = _ffi.DeviceArray.from_host(xr); yd2 = _ffi.DeviceArray(xr.size, np.float32)
zi = rng.standard_normal((8, 2))
iir.filter_local_dev(xd2, yd2, zi=zi)
want = signal.sosfilt(sos, xr.astype(np.float64), zi=zi)[0]
assert np.max(np.abs(yd2.to_host() - want)) / np.max(np.abs(want)) < 1e-6, 'sharded IIR via RCCL transport'
# fused sharded FIR: halo on the comm stream, tiles 1.. on the compute stream, tile 0 after the halo
from oracle import oracle as orc
bl = np.hanning(1024) / 512
xs = (rng.standard_normal(200000) + 1j * rng.standard_normal(200000)).astype(np.complex64)
fir = sharding.ShardedFIR(bl, tr, dtype=np.complex64)
xsd = fir.new_shard_buffer(xs.size); xsd.write(xs)
ysd = _ffi.DeviceArray(xs.size, np.complex64)
for _ in range(3):
    fir.filter_local_dev(xsd, ysd)
_ffi.sync()
ref = orc.fir_filter(bl, xs)
assert np.max(np.abs(ysd.to_host() - ref)) / np.max(np.abs(ref)) < 1e-6, 'overlapped shard filter'
# the same launch with a REAL halo: the one rank sends its tail to itself (option shard_self_halo), so the RCCL
# send/recv kernel runs beside the persistent filter launch, a one-thread kernel behind it bumps the device flag, and
# the workgroup that owns tile 0 (walked last) waits for that flag before it reads the 1023 samples in front of x.
# The tail is rewritten before every step: a stale L1 / L2 line of the previous halo would show up in y[0:1023].
_ffi.set_option('shard_self_halo', 1)
ns = 1 << 22
xs = (rng.standard_normal(ns) + 1j * rng.standard_normal(ns)).astype(np.complex64)
xsd = fir.new_shard_buffer(ns); xsd.write(xs)
ysd = _ffi.DeviceArray(ns, np.complex64)
for it in range(12):
    tail = (rng.standard_normal(1023) + 1j * rng.standard_normal(1023)).astype(np.complex64)
    xsd.write(tail, at=ns - 1023)
    xs[ns - 1023:] = tail
    _ffi.set_option('shard_two_launches', it %% 4 == 3)   # every fourth step through the two-launch form
    fir.filter_local_dev(xsd, ysd)
    got = ysd.to_host(0, 9000)
    ref = orc.fir_filter(bl, xs[:9000], hist=tail)
    e = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    assert e < 1e-6, ('self-halo step', it, e)
    got = ysd.to_host(ns - 9000, 9000)
    ref = orc.fir_filter(bl, xs[ns - 9000 - 1023:])[1023:]
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 1e-6, ('self-halo tail', it)
_ffi.set_option('shard_two_launches', 0)
import time
def step_ms(k=200):
    for _ in range(50): fir.filter_local_dev(xsd, ysd)
    _ffi.sync(); _ffi.timer_start()
    for _ in range(k): fir.filter_local_dev(xsd, ysd)
    return _ffi.timer_stop() / k
t1 = step_ms(); _ffi.set_option('shard_two_launches', 1); t2 = step_ms(); _ffi.set_option('shard_two_launches', 0)
print('self-halo 2^22 step: flag-in-kernel %%.4f ms, two launches %%.4f ms' %% (t1, t2))
_ffi.set_option('shard_self_halo', 0)
_ffi.check(L.skdsp_dist_shutdown())
print('RCCL_OK')
""" % ((os.path.dirname(os.path.dirname(os.path.abspath(__file__))),) * 2)
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert b"RCCL_OK" in out.stdout, out.stdout.decode()[-3000:]
    print(out.stdout.decode()[-300:])


@pytest.mark.parametrize("probe", [1, 2])
def test_sharded_step_earns_the_overlapped_form(probe):
    """dist.hip's probation: a process's first sharded step runs two launches, its second the overlapped launch on probation (short
    poll, tile 0 repeated behind the halo event, one sync), and only a passed probation (state 2) makes the overlapped form the steady
    state; a failed one (probe = 2: the test hook never publishes the probation step's flag) switches to two launches for good
    (state 3) -- and EVERY step's result is right whatever the poll did: a first-ever multi-GPU run loses neither seconds nor a step."""
    import subprocess
    import sys
    code = r"""
import os, sys, ctypes, time, numpy as np
sys.path.insert(0, os.path.join(%r, 'scikit-dsp-comm_amd'))
sys.path.insert(0, %r)
os.environ['SKDSP_DIST_FORCE_COMM'] = '1'
os.environ['SKDSP_SHARD_SELF_HALO'] = '1'      # the one rank is its own left neighbour: a REAL halo from the first step on
os.environ['SKDSP_SHARD_PROBE'] = '%d'
from sk_dsp_comm_amd import _ffi, sharding
from oracle import oracle as orc
_ffi.init(0)
L = _ffi.load()
buf = ctypes.create_string_buffer(128)
_ffi.check(L.skdsp_dist_unique_id(buf))
_ffi.check(L.skdsp_dist_init(0, 1, ctypes.c_char_p(buf.raw)))
tr = sharding.RcclTransport.__new__(sharding.RcclTransport)
tr.rank, tr.world, tr._rdzv = 0, 1, None
rng = np.random.default_rng(3)
bl = np.hanning(1024) / 512
ns = 1 << 21
xs = (rng.standard_normal(ns) + 1j * rng.standard_normal(ns)).astype(np.complex64)
fir = sharding.ShardedFIR(bl, tr, dtype=np.complex64)
xsd = fir.new_shard_buffer(ns); xsd.write(xs)
ysd = _ffi.DeviceArray(ns, np.complex64)
assert _ffi.get_option('shard_halo_state') == 0
states, walls = [], []
for it in range(5):
    tail = (rng.standard_normal(1023) + 1j * rng.standard_normal(1023)).astype(np.complex64)
    xsd.write(tail, at=ns - 1023)
    xs[ns - 1023:] = tail
    _ffi.sync()
    t0 = time.perf_counter()
    fir.filter_local_dev(xsd, ysd)
    _ffi.sync()
    walls.append(time.perf_counter() - t0)
    states.append((_ffi.get_option('shard_halo_state'), _ffi.get_option('shard_two_launches')))
    got = ysd.to_host(0, 9000)
    ref = orc.fir_filter(bl, xs[:9000], hist=tail)
    e = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    assert e < 1e-6, ('head (consumes the halo)', it, e, states)
    got = ysd.to_host(ns - 9000, 9000)
    ref = orc.fir_filter(bl, xs[ns - 9000 - 1023:])[1023:]
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 1e-6, ('tail', it)
print('states', states, 'wall ms', [round(w * 1e3, 2) for w in walls])
want = [(1, 0), (2, 0), (2, 0), (2, 0), (2, 0)] if %d == 1 else [(1, 0), (3, 1), (3, 1), (3, 1), (3, 1)]
assert states == want, states
assert max(walls[1:]) < 0.25, walls      # the failed probation costs milliseconds, not the seconds of the steady-state poll bound
_ffi.check(L.skdsp_dist_shutdown())
print('PROBATION_OK')
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), probe, probe)
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert b"PROBATION_OK" in out.stdout, out.stdout.decode()[-3000:]
    print(out.stdout.decode()[-400:])


def test_sharded_fir_eight_shards_emulated_on_one_gpu():
    """BASELINE config 5 semantics on one GPU: the 8-way sample-block sharding of
    sk_dsp_comm_amd.sharding (shard_bounds, headroom layout, Ntaps-1 halo in front of the
    shard, n_hist) with the halo hop done by a device copy instead of RCCL.  The sharded
    outputs must equal the single-vector filter (spot-checked against the oracle too)."""
    import ctypes
    from sk_dsp_comm_amd import sharding
    g = load("g5_fir1024.npz")
    b = g["b"]
    P = len(b)
    n, world = 2 ** 23 + 12345, 8
    L = _ffi.load()
    k = _ffi.FirKernel(b, _ffi.C64)
    xd = _ffi.DeviceArray(n, np.complex64).fill_noise(99)
    yd = _ffi.DeviceArray(n, np.complex64)
    k.filter_dev(xd, yd)
    y_full = yd.to_host()
    bounds = sharding.shard_bounds(n, world)
    y_parts = []
    for r, (s0, s1) in enumerate(bounds):
        nl = s1 - s0
        sh = _ffi.DeviceArray(nl, np.complex64, headroom=P - 1)
        _ffi.check(L.skdsp_memcpy_d2d(ctypes.c_void_p(sh.ptr), ctypes.c_void_p(xd.ptr + s0 * 8), nl * 8))
        if r == 0:
            _ffi.check(L.skdsp_dist_halo_exchange(ctypes.c_void_p(sh.ptr), nl, P - 1, _ffi.C64))  # world 1: zero fill
        else:  # what rank r-1 would send: its last P-1 samples
            _ffi.check(L.skdsp_memcpy_d2d(ctypes.c_void_p(sh.ptr - (P - 1) * 8), ctypes.c_void_p(xd.ptr + (s0 - (P - 1)) * 8), (P - 1) * 8))
        yo = _ffi.DeviceArray(nl, np.complex64)
        k.filter_dev(sh, yo, n_hist=P - 1)
        y_parts.append(yo.to_host())
    y_sh = np.concatenate(y_parts)
    assert y_sh.shape == y_full.shape
    assert_close(y_sh, y_full, 5e-7, "sharded vs single vector")
    for r in (1, 4, 7):  # oracle across a shard boundary
        s0 = bounds[r][0]
        xs = xd.to_host(s0 - 3000, 6000)
        ref = orc.fir_filter(b, xs)[P - 1:]
        assert_close(y_sh[s0 - 3000 + P - 1:s0 + 3000], ref, TOL32, "boundary of shard %d" % r)


class _CopyTransport:
    """Stands in for RcclTransport on one GPU: the halo hop is a device copy out of the
    full vector (what rank r-1 would have sent), rank 0 gets zeros."""

    def __init__(self, rank, world, full_xd, start):
        self.rank, self.world, self.full, self.start = rank, world, full_xd, start

    def halo_exchange_dev(self, xd, n, n_halo):
        import ctypes
        L = _ffi.load()
        esz = xd.dtype.itemsize
        if n_halo == 0:
            return
        if self.rank == 0:
            _ffi.check(L.skdsp_memset(ctypes.c_void_p(xd.ptr - n_halo * esz), 0, n_halo * esz))
        else:
            _ffi.check(L.skdsp_memcpy_d2d(ctypes.c_void_p(xd.ptr - n_halo * esz),
                                          ctypes.c_void_p(self.full.ptr + (self.start - n_halo) * esz), n_halo * esz))


@pytest.mark.parametrize("mode,L,M", [("filter", 1, 1), ("up", 4, 1), ("dn", 1, 3), ("updn", 4, 3), ("up", 12, 1), ("dn", 1, 12)])
def test_sharded_fir_driver_all_modes_emulated(mode, L, M):
    """sharding.ShardedFIR (.filter/.up/.dn/updn shards, halo lengths, shard alignment) against
    the single-vector result, 5 ranks emulated on one GPU."""
    import ctypes
    import math
    from sk_dsp_comm_amd import sharding
    g = load("g6_fir512_updn.npz")
    b = g["b"]
    n, world = 600_011, 5
    lib = _ffi.load()
    xd = _ffi.DeviceArray(n, np.complex64).fill_noise(5)
    k = _ffi.FirKernel(b, _ffi.C64)
    n_out = n if mode == "filter" else (n * L) // M
    yd = _ffi.DeviceArray(n_out, np.complex64)
    if mode == "filter":
        k.filter_dev(xd, yd)
    else:
        k.updn_dev(xd, yd, L, M)
    y_full = yd.to_host()
    mult = M // math.gcd(L, M)
    bounds = sharding.shard_bounds(n, world, multiple=mult)
    parts = []
    for r, (s0, s1) in enumerate(bounds):
        nl = s1 - s0
        fir = sharding.ShardedFIR(b, _CopyTransport(r, world, xd, s0), dtype=np.complex64)
        halo = fir.halo if mode in ("filter", "dn") else fir.up_halo(L)
        sh = _ffi.DeviceArray(nl, np.complex64, headroom=halo)
        _ffi.check(lib.skdsp_memcpy_d2d(ctypes.c_void_p(sh.ptr), ctypes.c_void_p(xd.ptr + s0 * 8), nl * 8))
        nlo = nl if mode == "filter" else (nl * L) // M
        yo = _ffi.DeviceArray(max(nlo, 1), np.complex64)
        if mode == "filter":
            fir.filter_local_dev(sh, yo)
        elif mode == "up":
            fir.up_local_dev(sh, yo, L)
        elif mode == "dn":
            fir.dn_local_dev(sh, yo, M)
        else:
            fir.updn_local_dev(sh, yo, L, M)
        parts.append(yo.to_host(0, nlo))
    y_sh = np.concatenate(parts)
    assert y_sh.shape == y_full.shape, (y_sh.shape, y_full.shape)
    assert_close(y_sh, y_full, 5e-7, "sharded %s L=%d M=%d" % (mode, L, M))


# ---------------------------------------------------------------------------
# block streaming: state in / state out
# ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64, np.complex128])
def test_iir_stream_matches_sosfilt_zi(dt):
    """filter_stream(x, zi) == scipy sosfilt(sos, x, zi=zi), and block-wise == one shot."""
    from scipy import signal
    import sk_dsp_comm_amd.multirate_helper as mrh
    rng = np.random.default_rng(41)
    sos = signal.butter(8, 0.2, output="sos")
    cplx = np.dtype(dt).kind == "c"
    n = 300_000
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    x = x.astype(dt)
    zi = rng.standard_normal((4, 2)) + (1j * rng.standard_normal((4, 2)) if cplx else 0)
    y_ref, zf_ref = signal.sosfilt(sos, x.astype(np.complex128 if cplx else np.float64), zi=zi)
    f = mrh.multirate_IIR(sos)
    y, zf = f.filter_stream(x, zi)
    tol = 1e-6 if np.dtype(dt).itemsize in (4, 8) and dt in (np.float32, np.complex64) else 1e-11
    assert max(rel_err(y, y_ref)) <= tol
    assert max(rel_err(zf, zf_ref)) <= tol
    # ragged blocks, including one shorter than a chunk and an empty one
    cuts = [0, 17, 17, 5000, 130_001, n]
    zs, outs = None, []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        yb, zs = f.filter_stream(x[lo:hi], zs)
        outs.append(yb)
    y_blocks = np.concatenate(outs)
    y_one = f.filter(x)
    assert max(rel_err(y_blocks, y_one)) <= tol
    assert max(rel_err(zs, signal.sosfilt(sos, x.astype(np.complex128 if cplx else np.float64),
                                      zi=np.zeros((4, 2)))[1])) <= tol


@pytest.mark.gpu
def test_iir_stream_slow_decay_scan_path():
    """A pole at radius 0.99999 keeps the workgroup-aggregate scan (K2) in play: zi must ride it."""
    from scipy import signal
    import sk_dsp_comm_amd.multirate_helper as mrh
    r = 0.99999
    sos = np.array([[1.0, 0.5, 0.0, 1.0, -2 * r * np.cos(0.3), r * r]])
    rng = np.random.default_rng(42)
    x = rng.standard_normal(1 << 20)
    zi = np.array([[0.7, -0.3]])
    y_ref, zf_ref = signal.sosfilt(sos, x, zi=zi)
    y, zf = mrh.multirate_IIR(sos).filter_stream(x, zi)
    assert max(rel_err(y, y_ref)) <= 1e-9
    assert max(rel_err(zf, zf_ref)) <= 1e-9


@pytest.mark.gpu
def test_iir_stream_long_cascade_split():
    from scipy import signal
    import sk_dsp_comm_amd.multirate_helper as mrh
    sos = signal.butter(30, 0.3, output="sos")  # 15 sections -> two chained device cascades
    rng = np.random.default_rng(43)
    x = rng.standard_normal(50_000)
    zi = 0.1 * rng.standard_normal((15, 2))
    y_ref, zf_ref = signal.sosfilt(sos, x, zi=zi)
    y, zf = mrh.multirate_IIR(sos).filter_stream(x, zi)
    assert max(rel_err(y, y_ref)) <= 1e-9
    assert max(rel_err(zf, zf_ref)) <= 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("dt,ntaps", [(np.complex64, 1024), (np.float32, 127), (np.float64, 33)])
def test_fir_stream_blocks_equal_one_shot(dt, ntaps):
    import sk_dsp_comm_amd.multirate_helper as mrh
    rng = np.random.default_rng(44)
    b = rng.standard_normal(ntaps) / ntaps
    n = 200_000
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if np.dtype(dt).kind == "c" else 0)
    x = x.astype(dt)
    f = mrh.multirate_FIR(b)
    y_one = f.filter(x)
    cuts = [0, 5, 5, 3000, 70_001, n]
    zs, outs = None, []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        yb, zs = f.filter_stream(x[lo:hi], zs)
        outs.append(yb)
    tol = 1e-6 if np.dtype(dt).itemsize <= 8 and dt != np.float64 else 1e-12
    assert max(rel_err(np.concatenate(outs), y_one)) <= tol
    assert np.array_equal(zs, x[n - (ntaps - 1):])


class _StateMailbox:
    """Emulated all-gather for ShardedIIR: ranks run one after the other in this process; rank r
    only folds rows k < r, which the earlier ranks of the sweep have already contributed."""
    def __init__(self, world):
        self.world, self.rank, self.rows = world, 0, {}

    def allgather_state(self, vec):
        self.rows[self.rank] = np.array(vec, copy=True)
        return np.stack([self.rows.get(k, np.zeros_like(vec)) for k in range(self.world)])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64])
@pytest.mark.parametrize("path", ["host", "dev"])
def test_sharded_iir_emulated_ranks(dt, path):
    """5 ragged shards, exact state hand-off, HIP scan per shard == sosfilt of the whole vector."""
    from scipy import signal
    from sk_dsp_comm_amd import sharding, _ffi
    rng = np.random.default_rng(51)
    n = 1_000_003
    cplx = np.dtype(dt).kind == "c"
    x = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dt)
    tol = 1e-6 if np.dtype(dt).name in ("float32", "complex64") else 1e-10
    for sos in (signal.ellip(8, 0.5, 60, [0.2, 0.4], btype="bandpass", output="sos"),  # 8 biquads (config 4)
                np.array([[1.0, 0.0, 0.0, 1.0, -0.999999, 0.0]])):                   # near-integrator: full re-filter
        want = signal.sosfilt(sos, x.astype(np.complex128 if cplx else np.float64))
        tr = _StateMailbox(5)
        got = []
        for r, (a, b) in enumerate(sharding.shard_bounds(n, 5)):
            tr.rank = r
            iir = sharding.ShardedIIR(sos, tr, dtype=dt)
            if path == "host":
                got.append(iir.filter_local_host(x[a:b]))
            else:
                xd = _ffi.DeviceArray.from_host(x[a:b])
                yd = _ffi.DeviceArray(b - a, dt)
                iir.filter_local_dev(xd, yd)
                got.append(yd.to_host())
                xd.free()
                yd.free()
        assert max(rel_err(np.concatenate(got), want)) <= tol


# ---------------------------------------------------------------------------------------------
# callers around the hot path against vectors captured from the reference (G11)
# ---------------------------------------------------------------------------------------------
def _g11():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_callers.npz"))


@pytest.mark.gpu
def test_interp24_deci24_match_reference():
    from sk_dsp_comm_amd import sigsys, config
    g = _g11()
    old = config.strict_dtype
    config.strict_dtype = True
    try:
        y = sigsys.interp24(g["i24_x"])
        assert y.dtype == np.float64 and max(rel_err(y, g["i24_y"])) <= 1e-9
        yc = sigsys.interp24(g["i24c_x"])
        assert yc.dtype == np.complex128 and max(rel_err(yc, g["i24c_y"])) <= 1e-9
        d = sigsys.deci24(g["d24_x"])
        assert max(rel_err(d, g["d24_y"])) <= 1e-9
        y32 = sigsys.interp24(g["i24_x"].astype(np.float32))  # float32 signal: float32 I/O kernels
        assert y32.dtype == np.float64 and max(rel_err(y32, g["i24_y"])) <= 2e-6
    finally:
        config.strict_dtype = old


@pytest.mark.gpu
def test_ten_band_eq_matches_reference():
    from sk_dsp_comm_amd import sigsys
    g = _g11()
    assert max(rel_err(sigsys.ten_band_eq_filt(g["eq_x"], g["eq_gdb"]), g["eq_y"])) <= 1e-9
    assert max(rel_err(sigsys.ten_band_eq_filt(g["eq_x"], g["eq_gdb"], Q=2.0), g["eq_y_q2"])) <= 1e-9
    with pytest.raises(ValueError):
        sigsys.ten_band_eq_filt(g["eq_x"], g["eq_gdb"][:9])


@pytest.mark.gpu
@pytest.mark.parametrize("pulse", ["rect", "rc", "src"])
def test_nrz_bits2_matches_reference(pulse):
    from sk_dsp_comm_amd import sigsys
    g = _g11()
    x, b = sigsys.nrz_bits2(g["nrz_bits"], 10, pulse, 0.25, 6)
    assert np.allclose(b, g["nrz_b_" + pulse], rtol=1e-13, atol=1e-16)
    assert x.shape == g["nrz_x_" + pulse].shape and max(rel_err(x, g["nrz_x_" + pulse])) <= 1e-11


@pytest.mark.gpu
def test_gray_transmitters_match_reference():
    from sk_dsp_comm_amd import digitalcom as dc
    g = _g11()
    data = g["tx_data"]
    for mod in (2, 4, 16, 64, 256):
        for pulse, ns in (("src", 8), ("rect", 4)):
            x, b, d = dc.qam_gray_encode_bb(None, ns, mod, pulse, 0.35, 6, data)
            ref = g["qam%d_%s_x" % (mod, pulse)]
            assert x.shape == ref.shape and max(rel_err(x, ref)) <= 1e-11, (mod, pulse)
            assert np.allclose(b, g["qam%d_%s_b" % (mod, pulse)], rtol=1e-13)
    for mod in (2, 4, 8, 16, 32):
        x, b, d = dc.mpsk_gray_encode_bb(None, 8, mod, "src", 0.25, 6, data)
        assert max(rel_err(x, g["mpsk%d_x" % mod])) <= 1e-11, mod
        assert np.allclose(b, g["mpsk%d_b" % mod], rtol=1e-13)
    x, b, d = dc.mpsk_gray_encode_bb(None, 5, 8, "rc", 0.35, 4, data)
    assert max(rel_err(x, g["mpsk8_rc_x"])) <= 1e-11


@pytest.mark.gpu
def test_fft_filt_bank_matches_reference(capsys):
    from sk_dsp_comm_amd import sigsys
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_filtbank.npz"))
    y, f, fd = sigsys.fft_filt_bank(g["xr"], g["h"] + 0j, n_fft2=128, n_bands2=2, bs=200, fs=1000)
    assert y.shape == g["odd_y"].shape and y.dtype == np.complex128
    assert max(rel_err(y, g["odd_y"])) <= 1e-11 and np.all(y[:, 2944:] == 0)
    assert np.allclose(f, g["odd_f"]) and np.allclose(fd, g["odd_fd"])
    y, f, fd = sigsys.fft_filt_bank(g["xr"], g["h"] + 0j, n_fft2=128, n_bands2=2, bs=200, fs=1000, n_band_odd=False)
    assert max(rel_err(y, g["even_y"])) <= 1e-11 and np.allclose(f, g["even_f"]) and np.allclose(fd, g["even_fd"])
    y, f, fd = sigsys.fft_filt_bank(g["xc"], g["hc"], n_fft2=100, n_bands2=1, bs=130, fs=1000)
    assert max(rel_err(y, g["cplx_y"])) <= 1e-11 and np.allclose(f, g["cplx_f"]) and np.allclose(fd, g["cplx_fd"])
    assert capsys.readouterr().out == str(g["stdout"])
    y32, _, _ = sigsys.fft_filt_bank(g["xr"].astype(np.float32), g["h"], n_fft2=128, n_bands2=2, bs=200, fs=1000)
    assert max(rel_err(y32, g["odd_y"])) <= 2e-6
    with pytest.raises(ValueError):
        sigsys.fft_filt_bank(g["xr"], g["h"], n_fft2=32)


@pytest.mark.gpu
def test_header_taps_drive_the_gpu_filter(tmp_path):
    """coeff2header round trip into the filter objects (8f-4): Q15 header -> multirate_FIR, SOS header -> multirate_IIR."""
    from scipy import signal
    from sk_dsp_comm_amd import coeff2header as c2h
    import sk_dsp_comm_amd.multirate_helper as mrh
    rng = np.random.default_rng(77)
    x = rng.standard_normal(20000)
    h = signal.firwin(64, 0.25)
    fn = str(tmp_path / "h.h")
    c2h.fir_fix_header(fn, h)
    hq = c2h.read_fir_header(fn)
    assert max(rel_err(mrh.multirate_FIR(hq).filter(x), signal.lfilter(hq, 1, x))) <= 1e-11
    sos = signal.ellip(6, 0.5, 60, 0.3, output="sos")
    fs = str(tmp_path / "s.h")
    c2h.iir_sos_header(fs, sos)
    sos_r = c2h.read_sos_header(fs)
    assert max(rel_err(mrh.multirate_IIR(sos_r).filter(x), signal.sosfilt(sos_r, x))) <= 1e-9


@pytest.mark.parametrize("L,M", [(2, 1), (3, 1), (4, 1), (3, 2), (2, 3), (4, 5), (4, 3), (5, 3)])
def test_c64_polyphase_large_tiles(L, M):
    """complex64 x real taps at sizes where the R=8 sliding-window tiles (and for Lp = 2..4 the
    unrolled-class kernel with full-row interleaved stores) are selected: windows against the oracle,
    including the very first and the very last outputs."""
    g = load("g6_fir512_updn.npz")
    b = g["b"]
    n = 3 * 2 ** 20 + 1234  # ragged: last workgroup and last rows are partial
    n -= n % M
    k = _ffi.FirKernel(b, _ffi.C64)
    xd = _ffi.DeviceArray(n, np.complex64).fill_noise(23)
    n_out = (n * L) // M
    yd = _ffi.DeviceArray(n_out, np.complex64)
    k.updn_dev(xd, yd, L, M)
    _ffi.sync()
    q = M // int(np.gcd(L, M))
    for s_in in (0, 700_000, n - 5000):
        s_in -= s_in % q
        lo = max(0, s_in - 600)
        lo -= lo % M
        xs = xd.to_host(lo, min(5000, n - lo))
        ref = orc.downsample(orc.fir_up(b, xs, L), M) if M > 1 else orc.fir_up(b, xs, L)
        m0 = (s_in * L) // M
        skip = ((s_in - lo) * L) // M
        w = min(len(ref) - skip, n_out - m0, 4000)
        assert w > 1000
        assert_close(yd.to_host(m0, w), ref[skip:skip + w], TOL32, "L=%d M=%d window @%d" % (L, M, s_in))


# ---------------------------------------------------------------------------------------------
# C-ABI error behaviour on a live device: status codes + messages, no crashes, no side effects
# ---------------------------------------------------------------------------------------------
def test_c_abi_rejects_bad_arguments():
    import ctypes
    L = _ffi.load()
    vp = ctypes.c_void_p
    b = np.ones(8)
    fir = _ffi.FirKernel(b, _ffi.C64)
    iir = _ffi.IirKernel(_ffi.F32, sos=np.array([[1.0, 0, 0, 1, -0.5, 0]]))
    x = _ffi.DeviceArray(1024, np.complex64).fill_noise(1)
    y = _ffi.DeviceArray(4096, np.complex64)

    def code(rc):
        assert rc != 0
        msg = L.skdsp_last_error().decode()
        assert msg, "an error status must come with a message"
        return rc

    # wrong handle kind, factors < 1, null handle
    assert code(L.skdsp_fir_filter_dev(vp(iir.h), vp(x.ptr), 1024, 0, vp(y.ptr))) == -1
    assert code(L.skdsp_iir_filter_dev(vp(fir.h), vp(x.ptr), 1024, vp(y.ptr))) == -1
    assert code(L.skdsp_fir_up_dev(vp(fir.h), vp(x.ptr), 1024, 0, 0, vp(y.ptr))) == -1
    assert code(L.skdsp_fir_dn_dev(vp(fir.h), vp(x.ptr), 1024, 0, 0, vp(y.ptr))) == -1
    assert code(L.skdsp_fir_filter_dev(vp(0), vp(x.ptr), 1024, 0, vp(y.ptr))) == -1
    # creation: no taps, bad dtype, complex taps for a real signal, sos without a0 == 1
    h = vp(0)
    assert code(L.skdsp_fir_create(_ffi._ptr(b), 0, 0, _ffi.C64, ctypes.byref(h))) == -1
    assert code(L.skdsp_fir_create(_ffi._ptr(b), 8, 0, 99, ctypes.byref(h))) == -1
    bad_sos = np.array([[1.0, 0, 0, 2.0, -0.5, 0]])
    assert code(L.skdsp_sos_create(_ffi._ptr(bad_sos), 1, _ffi.F32, ctypes.byref(h))) == -1
    bc = np.ones(4, dtype=np.complex128)
    rc = L.skdsp_fir_create(_ffi._ptr(bc), 4, 1, _ffi.F32, ctypes.byref(h))
    if rc == 0:  # creation may defer the check to the first launch
        xr = _ffi.DeviceArray(64, np.float32).fill_noise(2)
        yr = _ffi.DeviceArray(64, np.float32)
        assert code(L.skdsp_fir_filter_dev(h, vp(xr.ptr), 64, 0, vp(yr.ptr))) == -1
        L.skdsp_destroy(h)
    else:
        assert rc == -1
    # zero-length work is a no-op, not an error
    assert L.skdsp_fir_filter_dev(vp(fir.h), vp(x.ptr), 0, 0, vp(y.ptr)) == 0
    assert L.skdsp_iir_filter_dev(vp(iir.h), vp(x.ptr), 0, vp(y.ptr)) == 0
    assert L.skdsp_upsample_dev(vp(x.ptr), 0, 3, _ffi.C64, ctypes.c_double(1.0), vp(y.ptr)) == 0
    # the library still works after all of that
    fir.filter_dev(x, y, 1024)
    _ffi.sync()
    ref = orc.fir_filter(b, x.to_host())
    assert_close(y.to_host(0, 1024), ref, TOL32, "after errors")


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
@pytest.mark.parametrize("L,M", [(1, 1), (3, 1), (1, 4), (2, 3), (12, 1), (6, 4), (5, 2)])
def test_direct_fir_large_tiles_all_dtypes(dt, L, M):
    """Direct/polyphase kernels at sizes where the big sliding-window tiles and the vectorised
    interior staging (16-byte loads: 4 float32 / 2 complex64 or float64 / 1 complex128 per load) are
    taken, for every signal dtype; real and complex taps; windows at the start, middle and end."""
    rng = np.random.default_rng(61)
    n = 3 * 2 ** 20 + 777
    n -= n % M
    cplx = np.dtype(dt).kind == "c"
    tol = TOL32 if np.dtype(dt).itemsize // (2 if cplx else 1) == 4 else 1e-12
    for taps in (rng.standard_normal(33) / 6, (rng.standard_normal(21) + 1j * rng.standard_normal(21)) / 5):
        if np.iscomplexobj(taps) and not cplx:
            continue
        k = _ffi.FirKernel(taps, _ffi.code_of(dt))
        k.set_algo(_ffi.FIR_DIRECT)
        xd = _ffi.DeviceArray(n, dt).fill_noise(29)
        n_out = (n * L) // M
        yd = _ffi.DeviceArray(n_out, dt)
        k.updn_dev(xd, yd, L, M)
        _ffi.sync()
        q = M // int(np.gcd(L, M))
        for s_in in (0, 1_500_000, n - 4000):
            s_in -= s_in % q
            lo = max(0, s_in - 300)
            lo -= lo % M
            xs = xd.to_host(lo, min(4000, n - lo))
            ref = orc.fir_up(taps, xs, L)
            if M > 1:
                ref = orc.downsample(ref, M)
            m0 = (s_in * L) // M
            skip = ((s_in - lo) * L) // M
            w = min(len(ref) - skip, n_out - m0, 3000)
            assert w > 500
            assert_close(yd.to_host(m0, w), ref[skip:skip + w], tol, "%s L=%d M=%d @%d" % (np.dtype(dt).name, L, M, s_in))
        xd.free()
        yd.free()


@pytest.mark.parametrize("dt", [np.complex64, np.float32])
@pytest.mark.parametrize("M,ntaps", [(2, 512), (3, 512), (12, 768), (5, 1024), (7, 200), (4, 1024), (6, 300), (8, 2048), (10, 1500), (16, 4097), (24, 777),
                                     (32, 640), (48, 2000), (96, 3000)])
def test_fir_dn_overlap_save_decimating_store(M, ntaps, dt):
    """.dn of a long filter runs in the overlap-save engine: odd M through the decimating store (every full-rate output computed, every M-th
    kept), even M through the decimating INVERSE transform (the spectrum folded 2 / 4 / 8 / 16-fold between a thread's registers, what is left
    of M taken at the store) -- identical (to float32 rounding) to the direct polyphase kernel and to the decimating store, ragged length,
    with history; complex64 and the float32 two-real-tiles variant."""
    rng = np.random.default_rng(71)
    b = rng.standard_normal(ntaps) / np.sqrt(ntaps)
    n = 2 ** 20 + 12345
    esz = np.dtype(dt).itemsize
    k = _ffi.FirKernel(b, _ffi.code_of(dt))
    k.set_algo(_ffi.FIR_OLS)  # (left to itself, .dn prefers the matrix-pipe kernel where that one is faster)
    xd = _ffi.DeviceArray(n, dt, headroom=ntaps).fill_noise(31)
    xd.write(cnoise(rng, ntaps - 1) if dt == np.complex64 else rng.standard_normal(ntaps - 1).astype(np.float32), at=-(ntaps - 1))
    yd = _ffi.DeviceArray(n // M, dt)
    k.dn_dev(xd, yd, M, n_hist=ntaps - 1)
    y = yd.to_host()
    with _ffi.option("dn_no_ols", 1):
        y2 = _ffi.DeviceArray(n // M, dt)
        k.dn_dev(xd, y2, M, n_hist=ntaps - 1)
        y_direct = y2.to_host()
    assert_close(y, y_direct, 2e-6, "ols-dn vs direct M=%d" % M)
    if M % 2 == 0:   # the same call through the decimating store
        with _ffi.option("fir_dn_fold", 0):
            y3 = _ffi.DeviceArray(n // M, dt)
            k.dn_dev(xd, y3, M, n_hist=ntaps - 1)
            assert_close(y, y3.to_host(), 2e-6, "folded inverse vs decimating store M=%d" % M)
            y3.free()
    # and against the oracle on windows (incl. the history at the start and the ragged end)
    hist = np.empty(ntaps - 1, dt)
    import ctypes
    _ffi.check(_ffi.load().skdsp_memcpy_d2h(_ffi._ptr(hist), ctypes.c_void_p(xd.ptr - (ntaps - 1) * esz), hist.nbytes))
    head = np.concatenate([hist, xd.to_host(0, 6000)])
    ref = orc.fir_filter(b, head)[ntaps - 1:][::M]
    assert_close(y[:len(ref)], ref, TOL32, "ols-dn head M=%d" % M)
    lo = (n - 9000) - (n - 9000) % M
    tail = xd.to_host(lo - (ntaps - 1), n - lo + ntaps - 1)
    ref = orc.fir_filter(b, tail)[ntaps - 1:][::M][:n // M - lo // M]
    assert_close(y[lo // M:], ref, TOL32, "ols-dn tail M=%d" % M)


@pytest.mark.parametrize("dt", [np.complex64, np.complex128])
def test_iir_up_dn_complex_vs_scipy(dt):
    """Complex signals through the IIR interpolator / decimator: the zero-stuffed planes are built
    directly (no stuffed interleaved copy); against scipy's lfilter / sosfilt on the same data."""
    from scipy import signal
    import sk_dsp_comm_amd.multirate_helper as mrh
    rng = np.random.default_rng(81)
    n = 70_001
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dt)
    xw = x.astype(np.complex128)
    tol = 2e-6 if dt == np.complex64 else 1e-9
    for L in (1, 3, 12):
        rc = mrh.rate_change(L if L > 1 else 2)
        if L > 1:
            ref = signal.lfilter(rc.b, rc.a, L * np.kron(xw, np.r_[1.0, np.zeros(L - 1)]))
            assert max(rel_err(rc.up(x), ref)) <= tol, ("rate_change.up", L)
            ref_dn = signal.lfilter(rc.b, rc.a, xw)[::L][:n // L]
            assert max(rel_err(rc.dn(x), ref_dn)) <= tol, ("rate_change.dn", L)
    sos = signal.ellip(6, 0.5, 60, 0.2, output="sos")
    f = mrh.multirate_IIR(sos)
    ref = signal.sosfilt(sos, 5 * np.kron(xw, np.r_[1.0, np.zeros(4)]))
    assert max(rel_err(f.up(x, 5), ref)) <= tol
    assert max(rel_err(f.dn(x, 7), signal.sosfilt(sos, xw)[::7][:n // 7])) <= tol


@pytest.mark.parametrize("switch", ["SKDSP_FIR_MM", "SKDSP_FIR_BX"])
def test_direct_fir_kernels_behind_the_default_path(switch):
    """float32 / complex64 direct FIRs normally run as Toeplitz products of fp16 pieces on the matrix pipe
    (fir_bx.hip).  Behind it sit the FP32 / FP64 matrix-pipe kernels (fir_mm.hip: float64, long lag ranges;
    SKDSP_FIR_BX=0 forces them) and the register sliding-window kernels (complex128, complex taps;
    SKDSP_FIR_MM=0 forces them): both are re-checked here on the same float32 / complex64 cases."""
    import subprocess
    import sys
    env = dict(os.environ)
    env[switch] = "0"
    here = os.path.abspath(__file__)
    out = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k",
                          "c64_polyphase_large_tiles or fir_polyphase_all_ratios or g6_fir512 or updn_full_size"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = out.stdout.decode()[-600:]
    assert out.returncode == 0 and " passed" in tail, tail


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
@pytest.mark.parametrize("P,L,M", [(40, 1, 1), (301, 1, 1), (192, 12, 1), (192, 1, 12), (75, 5, 1), (96, 3, 2), (64, 2, 1),
                                   (33, 1, 3), (140, 7, 1), (90, 9, 4), (1024, 4, 3),
                                   # row tiles dealt to wave pairs (taps that do not fit one wave's registers)
                                   (512, 12, 1), (1000, 12, 1), (700, 8, 1), (1100, 8, 1), (500, 16, 1), (420, 12, 5), (333, 8, 3),
                                   # one row tile, lags dealt to the four waves (decimators with a large M)
                                   (512, 1, 12), (1000, 1, 12), (128, 1, 12), (300, 1, 16), (512, 1, 24), (77, 1, 20), (640, 1, 8),
                                   # ... with fewer slots per column where 16 of them do not fit the window
                                   (512, 1, 16), (1024, 1, 24), (256, 1, 32), (512, 1, 48), (200, 1, 100)])
def test_matrix_pipe_geometries(dt, P, L, M):
    """Row-tile / lag-block geometries of the matrix-pipe Toeplitz kernel (1..8 row tiles, 1..48 lag blocks, shapes it
    hands on to the kernels behind it), ragged lengths, history: against the oracle at the float32 tolerance."""
    rng = np.random.default_rng(P + 13 * L + M)
    b = rng.standard_normal(P) / np.sqrt(P)
    n = 700_003
    n -= n % M
    cplx = np.dtype(dt).kind == "c"
    x = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dt)
    k = _ffi.FirKernel(b, _ffi.code_of(dt))
    k.set_algo(_ffi.FIR_DIRECT)
    xd = _ffi.DeviceArray.from_host(x)
    yd = _ffi.DeviceArray(n * L // M, dt)
    k.updn_dev(xd, yd, L, M)
    _ffi.sync()
    got = yd.to_host()
    m = 60_000  # inputs checked at the head and at the tail
    ref = orc.fir_up(b, x[:m], L)
    if M > 1:
        ref = orc.downsample(ref, M)
    assert_close(got[:len(ref)], ref, TOL32, "head")
    lo = n - m
    lo -= lo % M
    pad = -(-P // L) + 2
    lo2 = lo - pad - ((lo - pad) % M)  # start early enough for the filter memory, on an output boundary
    ref = orc.fir_up(b, x[lo2:], L)
    if M > 1:
        ref = orc.downsample(ref, M)
    skip = (lo - lo2) * L // M
    assert_close(got[lo * L // M:], ref[skip:], TOL32, "tail")


class _View:
    """A window of a DeviceArray starting `off` samples in (misaligned device pointers on purpose)."""
    def __init__(self, base, off, n):
        self.ptr = base.ptr + off * base.dtype.itemsize
        self.n = n
        self.dtype = base.dtype
        self.code = base.code


@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64])
@pytest.mark.parametrize("L,M", [(1, 1), (4, 3), (3, 1)])
def test_direct_fir_misaligned_device_pointers(dt, L, M):
    """x and y that are only element-aligned (views into larger device arrays): the 16-byte fast paths
    of the matrix-pipe and sliding-window kernels must step aside."""
    rng = np.random.default_rng(91)
    b = rng.standard_normal(40) / 6
    n = 300_001
    n -= n % M
    big = _ffi.DeviceArray(n + 64, dt).fill_noise(37)
    ybig = _ffi.DeviceArray((n * L) // M + 64, dt)
    k = _ffi.FirKernel(b, _ffi.code_of(dt))
    k.set_algo(_ffi.FIR_DIRECT)
    for xo, yo in ((1, 0), (3, 1), (0, 1)):
        xv, yv = _View(big, xo, n), _View(ybig, yo, (n * L) // M)
        k.updn_dev(xv, yv, L, M, n=n)
        _ffi.sync()
        x = big.to_host(xo, 20000)
        ref = orc.fir_up(b, x, L)
        if M > 1:
            ref = orc.downsample(ref, M)
        got = ybig.to_host(yo, len(ref))
        tol = TOL32 if np.dtype(dt).itemsize // (2 if np.dtype(dt).kind == "c" else 1) == 4 else 1e-12
        assert_close(got, ref, tol, "%s L=%d M=%d offsets %d/%d" % (np.dtype(dt).name, L, M, xo, yo))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_iir_matrix_pipe_chunk_states_large(dt):
    """2^24 samples put the scan in its aggregate-free mode (chunk end states from the FP64 matrix pipe,
    carries straight from them) for both I/O precisions: head and deep windows against the oracle,
    and the same launch with the recurrence K1 (SKDSP_IIR_NO_MFMA) must agree to rounding."""
    g = load("g7_iir_sos.npz")
    sos = g["sos8"]
    n = 2 ** 24
    k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
    xd = _ffi.DeviceArray(n, dt).fill_noise(17)
    yd = _ffi.DeviceArray(n, dt)
    k.filter_dev(xd, yd)
    _ffi.sync()
    tol = TOL32 if dt == np.float32 else 1e-9
    m = 2 ** 19
    assert_close(yd.to_host(0, m), orc.sos_filter(sos, xd.to_host(0, m)), tol, "head")
    s0 = n - 2 * m
    ref = orc.sos_filter(sos, xd.to_host(s0 - 20000, m + 20000))[20000:]
    assert_close(yd.to_host(s0, m), ref, tol, "deep window")
    with _ffi.option("iir_no_mfma", 1):
        y2 = _ffi.DeviceArray(n, dt)
        k.filter_dev(xd, y2)
        _ffi.sync()
    assert_close(yd.to_host(s0, m), y2.to_host(s0, m), 1e-6 if dt == np.float32 else 1e-11, "matrix-pipe vs recurrence K1")


@pytest.mark.parametrize("nsec", [9, 12])
def test_iir_matrix_pipe_two_row_tiles(nsec):
    """9..12 biquads (18..24 states): the matrix-pipe K1 runs two 16-row tiles; against scipy."""
    from scipy import signal
    sos = signal.butter(2 * nsec, 0.35, output="sos")
    assert sos.shape[0] == nsec
    n = 2 ** 24
    k = _ffi.IirKernel(_ffi.F32, sos=sos)
    xd = _ffi.DeviceArray(n, np.float32).fill_noise(19)
    yd = _ffi.DeviceArray(n, np.float32)
    k.filter_dev(xd, yd)
    _ffi.sync()
    m = 2 ** 18
    assert_close(yd.to_host(0, m), signal.sosfilt(sos, xd.to_host(0, m).astype(np.float64)), TOL32, "head")
    s0 = n - 2 * m
    ref = signal.sosfilt(sos, xd.to_host(s0 - 30000, m + 30000).astype(np.float64))[30000:]
    assert_close(yd.to_host(s0, m), ref, TOL32, "deep window")


@pytest.mark.parametrize("dt,nsec", [(np.complex64, 8), (np.complex128, 8), (np.complex64, 3), (np.complex64, 12), (np.complex128, 10)])
def test_iir_complex_interleaved_kernels(dt, nsec):
    """Complex signals stay interleaved end to end in the aggregate-free mode (iir_k1c / iir_k3c): head, deep
    window and the streaming state against scipy, a ragged length, and agreement with the planar detour."""
    from scipy import signal
    sos = signal.butter(2 * nsec, 0.3, output="sos")
    D = 2 * nsec
    rng = np.random.default_rng(5)
    n = 2 ** 22 + 777
    k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
    xd = _ffi.DeviceArray(n, dt).fill_noise(23)
    yd = _ffi.DeviceArray(n, dt)
    zi = rng.standard_normal((nsec, 2)) + 1j * rng.standard_normal((nsec, 2))
    flat = np.concatenate([zi.real.ravel(), zi.imag.ravel()])  # C-ABI layout: [re plane | im plane]
    zf = k.filter_state_dev(xd, yd, zi=flat)
    _ffi.sync()
    zf = (zf[:D] + 1j * zf[D:]).reshape(nsec, 2)
    tol = TOL32 if dt == np.complex64 else 1e-10
    m = 2 ** 17
    ref, _ = signal.sosfilt(sos, xd.to_host(0, m).astype(np.complex128), zi=zi)
    assert_close(yd.to_host(0, m), ref, tol, "head")
    w = 40000
    ref, zf_ref = signal.sosfilt(sos, xd.to_host(n - m - w, m + w).astype(np.complex128), zi=np.zeros((nsec, 2), complex))
    assert_close(yd.to_host(n - m, m), ref[w:], tol, "tail window")
    assert_close(zf, zf_ref, tol, "final state")
    with _ffi.option("iir_planar", 1):
        y2 = _ffi.DeviceArray(n, dt)
        zf2 = k.filter_state_dev(xd, y2, zi=flat)
        _ffi.sync()
        zf2 = (zf2[:D] + 1j * zf2[D:]).reshape(nsec, 2)
    assert_close(yd.to_host(0, n), y2.to_host(0, n), 1e-6 if dt == np.complex64 else 1e-12, "interleaved vs planar")
    assert_close(zf, zf2, 1e-9, "state: interleaved vs planar")


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_direct_fir_random_geometries(seed):
    """Randomised (dtype, taps, L, M, length, history) sweep of the direct / polyphase path -- whichever kernel
    the dispatcher picks (fp16-piece or FP32 matrix pipe, sliding window) -- head and tail windows against the
    oracle at the float32 tolerance."""
    rng = np.random.default_rng(seed)
    for _ in range(40):
        L, M, P = int(rng.integers(1, 17)), int(rng.integers(1, 17)), int(rng.integers(1, 420))
        dt = [np.float32, np.complex64][int(rng.integers(0, 2))]
        n = int(rng.integers(1, 100_000)) if rng.random() < 0.3 else int(rng.integers(100_000, 900_000))
        n -= n % M
        if n == 0:
            continue
        hist = int(rng.integers(0, 2)) * (-(-(P - 1) // L))
        b = rng.standard_normal(P) / np.sqrt(P)
        cplx = np.dtype(dt).kind == "c"
        xa = (rng.standard_normal(n + hist) + (1j * rng.standard_normal(n + hist) if cplx else 0)).astype(dt)
        k = _ffi.FirKernel(b, _ffi.code_of(dt))
        k.set_algo(_ffi.FIR_DIRECT)
        xd = _ffi.DeviceArray(n, dt, headroom=max(hist, 1))
        xd.write(xa[hist:], at=0)
        if hist:
            xd.write(xa[:hist], at=-hist)
        yd = _ffi.DeviceArray(n * L // M, dt)
        k.updn_dev(xd, yd, L, M, n=n, n_hist=hist)
        _ffi.sync()
        got = yd.to_host()
        m = min(n, 20000)
        m -= m % M
        pad = -(-P // L) + 2
        for lo, hi in ((0, m), (n - m, n)):
            lo -= lo % M
            lo2 = max(lo - pad, -hist)
            ref = orc.fir_up(b, xa[hist + lo2: hist + hi], L)[(lo - lo2) * L::M]
            g = got[lo * L // M: lo * L // M + len(ref)]
            assert_close(g, ref[:len(g)], TOL32, "%s P=%d L=%d M=%d n=%d hist=%d @%d" % (np.dtype(dt).name, P, L, M, n, hist, lo))
        xd.free()
        yd.free()


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_matrix_pipe_path_is_scale_invariant(dt):
    """The matrix-pipe FIR path (fir_bx.hip) multiplies fp16 pieces, whose exponent range is narrow, so every window of the
    signal is scaled by the power of two that puts its largest magnitude at 2^14 and its outputs by the inverse: scaling the
    signal by 2^+-80 scales the output by exactly that factor, bit for bit, and a stretch far below the rest of its window keeps
    its own relative accuracy down to 2^-29 of the window's largest magnitude (fp16's floor under the lifted second piece)."""
    rng = np.random.default_rng(3)
    b = rng.standard_normal(96) / 10
    n = 400_000
    cplx = np.dtype(dt).kind == "c"
    x = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dt)
    x[1000] *= 1e4  # a loud sample next to quiet ones
    x[2000:2100] *= 1e-6
    x[5000:5100] *= 1e-4
    x[9000:9100] *= 1e-10
    k = _ffi.FirKernel(b, _ffi.code_of(dt))
    k.set_algo(_ffi.FIR_DIRECT)
    outs = []
    for e in (0, 80, -80):
        xd = _ffi.DeviceArray.from_host((x * np.float32(2.0) ** e).astype(dt))
        yd = _ffi.DeviceArray(n * 4 // 3 - 1, dt)
        k.updn_dev(xd, yd, 4, 3, n=n - n % 3)
        _ffi.sync()
        outs.append(yd.to_host())
        xd.free()
        yd.free()
    assert np.array_equal(outs[1], outs[0] * np.float32(2.0) ** 80)
    assert np.array_equal(outs[2], outs[0] * np.float32(2.0) ** -80)
    # the quiet stretches keep their own relative accuracy (error measured against THEIR level, not the window's): both fp16 pieces of a
    # sample are normal numbers down to 2^-29 of the window's largest magnitude (the second piece is carried lifted by 2^11) -- the float32
    # tolerance at 80 and 120 dB below the rest; 200 dB below, the first piece is subnormal and the second still holds 2^-12 of the sample
    ref = orc.downsample(orc.fir_up(b, x[:9300], 4), 3)
    for lo_in, hi_in, bound in ((5040, 5090, 1e-6), (2040, 2090, 1e-6), (9040, 9090, 2e-3)):
        lo, hi = lo_in * 4 // 3, hi_in * 4 // 3
        err = np.max(np.abs(outs[0][lo:hi] - ref[lo:hi])) / np.max(np.abs(ref[lo:hi]))
        assert err < bound, (lo_in, err)


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_matrix_pipe_dispatch_sweep(dt):
    """Every (taps, L, M) the direct path may be asked for must find a kernel -- the geometry test of the matrix-pipe kernel and the list of its
    instantiations are separate pieces of code (a geometry without a kernel raised NotImplementedError once) -- and hold the float32 tolerance:
    a sweep over tap counts x rate changes on short signals."""
    rng = np.random.default_rng(11)
    cplx = np.dtype(dt).kind == "c"
    worst = 0.0
    for P in (9, 40, 96, 130, 257, 400, 513, 700, 1025, 1500):
        b = rng.standard_normal(P) / np.sqrt(P)
        k = _ffi.FirKernel(b, _ffi.code_of(dt))
        k.set_algo(_ffi.FIR_DIRECT)
        for L, M in ((1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1), (12, 1), (16, 1), (24, 1), (1, 2), (1, 3), (1, 4), (1, 5), (1, 6),
                     (1, 8), (1, 12), (1, 16), (1, 24), (1, 32), (1, 48), (1, 100), (4, 3), (3, 2), (12, 5), (5, 12), (7, 4), (16, 3)):
            n = 30_000 - 30_000 % M
            x = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dt)
            xd = _ffi.DeviceArray.from_host(x)
            yd = _ffi.DeviceArray(n * L // M, dt)
            k.updn_dev(xd, yd, L, M)
            _ffi.sync()
            got = yd.to_host()
            ref = orc.downsample(orc.fir_up(b, x, L), M) if L > 1 else orc.fir_dn(b, x, M)
            e = np.max(np.abs(got - ref[:len(got)])) / np.max(np.abs(ref))
            worst = max(worst, e)
            assert e < TOL32, (P, L, M, e)
            xd.free(); yd.free()


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_matrix_pipe_impulse_and_silence(dt):
    """The matrix-pipe FIR on the two inputs that show its operand representation directly: an impulse returns the taps (two fp16 pieces each:
    2^-23 of the largest tap at worst), also where the taps span 200 dB; silence returns exact zeros (a window without a largest magnitude is
    not scaled); a window holding one tiny sample only is scaled by it."""
    rng = np.random.default_rng(7)
    cplx = np.dtype(dt).kind == "c"
    for P, L, M in ((127, 1, 1), (512, 12, 1), (512, 1, 12), (512, 4, 3)):
        b = rng.standard_normal(P) * np.logspace(0, -10, P)          # taps over 200 dB
        k = _ffi.FirKernel(b, _ffi.code_of(dt))
        k.set_algo(_ffi.FIR_DIRECT)
        n = 60_000 - 60_000 % M
        for amp in (1.0, 3e-33):
            x = np.zeros(n, dtype=dt)
            x[12345 - 12345 % M] = amp * ((0.6 - 0.8j) if cplx else 1.0)
            xd = _ffi.DeviceArray.from_host(x)
            yd = _ffi.DeviceArray(n * L // M, dt)
            k.updn_dev(xd, yd, L, M)
            _ffi.sync()
            got = yd.to_host()
            ref = orc.downsample(orc.fir_up(b, x, L), M) if L > 1 else orc.fir_dn(b, x, M)
            err = np.max(np.abs(got - ref[:len(got)])) / np.max(np.abs(ref))
            assert err < 3e-7, (P, L, M, amp, err)
            xd.free(); yd.free()
        xd = _ffi.DeviceArray.from_host(np.zeros(n, dtype=dt))
        yd = _ffi.DeviceArray(n * L // M, dt)
        k.updn_dev(xd, yd, L, M)
        _ffi.sync()
        assert not np.any(yd.to_host()), (P, L, M)
        xd.free(); yd.free()


@pytest.mark.parametrize("case", ["f32_direct_2p30", "c64_ols_2p29", "c64_updn43_2p29", "f32_ols_2p30"])
def test_large_index_ranges(case):
    """4-8 GiB signals (sample and byte offsets beyond 2^31/2^32): windows at the head, the middle and the tail of
    every FIR engine against the oracle."""
    import bench
    rng = np.random.default_rng(5)
    dt, n, b, L, M, algo = {
        "f32_direct_2p30": (np.float32, 1 << 30, rng.standard_normal(40) / 6, 1, 1, _ffi.FIR_DIRECT),
        "c64_ols_2p29": (np.complex64, 1 << 29, bench.firwin_lowpass(1024, 0.2), 1, 1, _ffi.FIR_OLS),
        "c64_updn43_2p29": (np.complex64, (1 << 29) - ((1 << 29) % 3), bench.firwin_lowpass(512, 0.225), 4, 3, _ffi.FIR_DIRECT),
        "f32_ols_2p30": (np.float32, 1 << 30, rng.standard_normal(300) / 17, 1, 1, _ffi.FIR_OLS),
    }[case]
    k = _ffi.FirKernel(b, _ffi.code_of(dt))
    k.set_algo(algo)
    xd = _ffi.DeviceArray(n, dt).fill_noise(9)
    n_out = n * L // M
    yd = _ffi.DeviceArray(n_out, dt)
    try:
        if L == 1 and M == 1:
            k.filter_dev(xd, yd)
        else:
            k.updn_dev(xd, yd, L, M)
        _ffi.sync()
        pad = -(-len(b) // L) + 2
        for lo in (0, (n // 2) - ((n // 2) % M), n - 30000 - ((n - 30000) % M)):
            hi = min(n, lo + 30000)
            lo2 = max(lo - pad, 0)
            ref = orc.fir_up(b, xd.to_host(lo2, hi - lo2), L)[(lo - lo2) * L::M]
            got = yd.to_host(lo * L // M, min(len(ref), n_out - lo * L // M))
            assert_close(got, ref[:len(got)], TOL32, "%s @%d" % (case, lo))
    finally:
        xd.free()
        yd.free()


@pytest.mark.parametrize("dt,lg", [(np.float32, 30), (np.complex64, 29), (np.float64, 29)])
def test_iir_large_index_ranges(dt, lg):
    """4 GiB signals through the IIR scan (8192-sample chunks, interleaved complex kernels): head, middle and tail
    windows against the oracle (the filter forgets after ~8k samples, so a window restarts 20000 samples early)."""
    import bench
    sos = bench.elliptic_bpf_sos()
    n = 1 << lg
    k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
    xd = _ffi.DeviceArray(n, dt).fill_noise(3)
    yd = _ffi.DeviceArray(n, dt)
    try:
        k.filter_dev(xd, yd)
        _ffi.sync()
        tol = TOL32 if np.dtype(dt).name in ("float32", "complex64") else 1e-9
        for lo in (0, n // 2, n - 60000):
            lo2 = max(lo - 20000, 0)
            ref = orc.sos_filter(sos, xd.to_host(lo2, lo + 40000 - lo2))[lo - lo2:]
            assert_close(yd.to_host(lo, 40000), ref, tol, "%s @%d" % (np.dtype(dt).name, lo))
    finally:
        xd.free()
        yd.free()


@pytest.mark.parametrize("dt", [np.float32, np.complex128])
def test_iir_unit_tail_factorisation_states(dt):
    """Cascades with b2 = b0 in every section run re-factored (all gain in section 0, b0 = b2 = 1 elsewhere); the
    states still cross the API in the caller's factorisation: wildly different per-section gains, zi in, zf out,
    block-wise == one shot, all against scipy.  A cascade that does not qualify (b2 != b0) takes the general form."""
    from scipy import signal
    import sk_dsp_comm_amd.multirate_helper as mrh
    rng = np.random.default_rng(17)
    sos = signal.ellip(10, 0.4, 70, 0.27, output="sos")
    g = np.array([3e3, 2e-4, -7.0, 1.0, 5e2])  # redistribute the gain between the sections (product kept)
    g[-1] = 1.0 / np.prod(g[:-1])
    sos[:, :3] *= g[:, None]
    assert np.allclose(sos[:, 2], sos[:, 0], rtol=1e-14)
    cplx = np.dtype(dt).kind == "c"
    n = 500_000
    x = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dt)
    zi = (rng.standard_normal((5, 2)) + (1j * rng.standard_normal((5, 2)) if cplx else 0)) * np.abs(g)[:, None]
    wide = np.complex128 if cplx else np.float64
    y_ref, zf_ref = signal.sosfilt(sos, x.astype(wide), zi=zi)
    f = mrh.multirate_IIR(sos)
    y, zf = f.filter_stream(x, zi)
    tol = 2e-6 if dt == np.float32 else 1e-10
    assert max(rel_err(y, y_ref)) <= tol
    assert np.max(np.abs(zf - zf_ref) / (np.abs(zf_ref).max(axis=1, keepdims=True) + 1e-300)) <= tol
    zs, outs = zi, []
    for lo, hi in ((0, 1000), (1000, 200_001), (200_001, n)):
        yb, zs = f.filter_stream(x[lo:hi], zs)
        outs.append(yb)
    assert max(rel_err(np.concatenate(outs), y_ref)) <= tol
    assert np.max(np.abs(zs - zf_ref) / (np.abs(zf_ref).max(axis=1, keepdims=True) + 1e-300)) <= tol
    sos2 = sos.copy()
    sos2[2, 2] *= 0.9  # no longer b2 == b0: general form
    y2, zf2 = mrh.multirate_IIR(sos2).filter_stream(x, zi)
    y2_ref, zf2_ref = signal.sosfilt(sos2, x.astype(wide), zi=zi)
    assert max(rel_err(y2, y2_ref)) <= tol
    assert np.max(np.abs(zf2 - zf2_ref) / (np.abs(zf2_ref).max(axis=1, keepdims=True) + 1e-300)) <= tol


@pytest.mark.parametrize("dt,n", [(np.float32, 2 ** 24), (np.float32, 2 ** 25 - 12345), (np.float32, 3 * 2 ** 24), (np.float32, 2 ** 26),
                                  (np.float64, 2 ** 25), (np.float32, 2 ** 26 + 2 ** 22)])
def test_iir_chunk_state_kernels_by_chunk_length(dt, n):
    """Chunk lengths 128 / 256 / 512 run the register-resident K1 (iir_k1r_kernel, 1 / 2 / 4 pieces per chunk, ragged
    last group), 384 and 640 the L2-streamed one: head, middle and tail windows against the oracle."""
    import bench
    sos = bench.elliptic_bpf_sos()
    k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
    xd = _ffi.DeviceArray(n, dt).fill_noise(5)
    yd = _ffi.DeviceArray(n, dt)
    try:
        k.filter_dev(xd, yd)
        _ffi.sync()
        tol = TOL32 if dt == np.float32 else 1e-9
        for lo in (0, n // 2 + 777, n - 50000):
            lo2 = max(lo - 20000, 0)
            ref = orc.sos_filter(sos, xd.to_host(lo2, lo + 40000 - lo2))[lo - lo2:]
            assert_close(yd.to_host(lo, 40000), ref, tol, "n=%d @%d" % (n, lo))
    finally:
        xd.free()
        yd.free()


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128])
@pytest.mark.parametrize("M,n", [(12, 2 ** 24), (3, 2 ** 24 - 1), (2, 3_000_001), (4, 2 ** 22), (7, 5_000_003), (5, 40_000), (4096, 2 ** 22)])
def test_iir_dn_decimating_store(dt, M, n):
    """.dn: K3 (real signals) / the interleaved complex K3 stores every M-th output itself.  Identical to the full-rate result followed by the
    downsample kernel (SKDSP_IIR_DN_FULL), and head / tail windows against the oracle; ragged lengths, M below and
    above the 4 samples of a staged segment."""
    import ctypes
    import bench
    sos = bench.elliptic_bpf_sos()
    k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
    xd = _ffi.DeviceArray(n, dt).fill_noise(2)
    yd = _ffi.DeviceArray(n // M + 8, dt)
    y2 = _ffi.DeviceArray(n // M + 8, dt)
    lib = _ffi.load()
    try:
        yd.write(np.full(n // M + 8, 7.0, dt), at=0)  # sentinel: nothing may be written beyond n // M outputs
        _ffi.check(lib.skdsp_iir_dn_dev(ctypes.c_void_p(k.h), ctypes.c_void_p(xd.ptr), n, M, ctypes.c_void_p(yd.ptr)))
        with _ffi.option("iir_dn_full", 1):
            _ffi.check(lib.skdsp_iir_dn_dev(ctypes.c_void_p(k.h), ctypes.c_void_p(xd.ptr), n, M, ctypes.c_void_p(y2.ptr)))
        _ffi.sync()
        got = yd.to_host(0, n // M + 8)
        assert np.all(got[n // M:] == 7.0)
        # (the same kernel with and without the decimating store: identical; when the two calls take different scan
        #  paths -- single-pass for the full-rate call -- they agree to float64 rounding instead)
        # (round 6: this cascade is admitted to the float32 from-rest states, and the two calls then run them over chunks of different lengths -- 96 and 128
        #  samples: two float32 sums of the same states, each within the contract; with the option off the two calls are bit-identical as before)
        v32 = dt in (np.float32, np.complex64) and _ffi.get_option("iir_par_v32") and _ffi.sos_par_info(sos)["v32_admitted"]
        assert_close(got[:n // M], y2.to_host(0, n // M), (1e-6 if v32 else 0.0) if dt in (np.float32, np.complex64) else 1e-12, "dn vs full-rate + downsample")
        if v32:
            with _ffi.option("iir_par_v32", 0):
                _ffi.check(lib.skdsp_iir_dn_dev(ctypes.c_void_p(k.h), ctypes.c_void_p(xd.ptr), n, M, ctypes.c_void_p(yd.ptr)))
                with _ffi.option("iir_dn_full", 1):
                    _ffi.check(lib.skdsp_iir_dn_dev(ctypes.c_void_p(k.h), ctypes.c_void_p(xd.ptr), n, M, ctypes.c_void_p(y2.ptr)))
            _ffi.sync()
            got = yd.to_host(0, n // M + 8)
            assert_close(got[:n // M], y2.to_host(0, n // M), 0.0, "dn vs full-rate + downsample, float64 from-rest states")
        tol = TOL32 if dt in (np.float32, np.complex64) else 1e-9
        m = min(n, 60000)
        ref = orc.sos_filter(sos, xd.to_host(0, m))[::M][:m // M]
        assert_close(got[:len(ref)], ref, tol, "head M=%d" % M)
    finally:
        xd.free()
        yd.free()
        y2.free()


# ----------------------------------------------------------------- single-pass IIR scan (iir_fused.hip)
@pytest.mark.parametrize("dt,n", [(np.float32, 9_000_017), (np.float32, 2 ** 24), (np.float64, 4_400_003), (np.complex64, 4_300_009),
                                  (np.complex128, 2_200_001)])
@pytest.mark.parametrize("filt", ["ellip8", "butter4", "biquad", "butter12", "cheby14"])
def test_iir_single_pass_scan_matches_two_pass_and_oracle(dt, n, filt):
    """The single-pass scan (one launch, x read once: segments of 256 register-resident chunks, from-rest scan,
    decoupled look-back, correction by binary powers) against the K1 / carries / K3 path (option iir_two_pass) and the
    oracle: ragged lengths, initial state in, final state out, repeated launches (look-back epochs).  Cascades of 6+ biquads
    (butter12, cheby14, ellip8) run their chunk scan on the matrix pipe: 2 .. 8 levels depending on filter and precision."""
    from scipy import signal
    import bench
    sos = {"ellip8": bench.elliptic_bpf_sos(), "butter4": signal.butter(4, 0.25, output="sos"),
           "biquad": signal.tf2sos(*signal.iirpeak(0.1, 30)), "butter12": signal.butter(12, 0.4, output="sos"),
           "cheby14": signal.cheby1(14, 0.5, 0.15, output="sos")}[filt]
    nsec = sos.shape[0]
    rng = np.random.default_rng(5)
    cplx = np.dtype(dt).kind == "c"
    single = dt in (np.float32, np.complex64)
    zi = rng.standard_normal((nsec, 2)) * 0.1 + (1j * rng.standard_normal((nsec, 2)) * 0.1 if cplx else 0)
    zi_flat = np.concatenate([zi.real.ravel(), zi.imag.ravel()]) if cplx else zi.ravel()   # C ABI: states of re, then of im
    k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
    xd = _ffi.DeviceArray(n, dt).fill_noise(77)
    y1 = _ffi.DeviceArray(n, dt)
    y2 = _ffi.DeviceArray(n, dt)
    D = 2 * nsec
    unflat = (lambda z: (z[:D] + 1j * z[D:])) if cplx else (lambda z: z)
    try:
        with _ffi.option("iir_two_pass", -1):  # (-1: single pass wherever it applies, also where the two-pass scan measured faster)
            zf1 = unflat(k.filter_state_dev(xd, y1, zi=zi_flat))
        with _ffi.option("iir_two_pass", 1):
            zf2 = unflat(k.filter_state_dev(xd, y2, zi=zi_flat))
        tol = TOL32 if single else 1e-12
        w = 1 << 20
        for s0 in (0, n // 2 - 12345, n - w):
            assert_close(y1.to_host(s0, w), y2.to_host(s0, w), tol, "single-pass vs two-pass @%d" % s0)
        assert_close(zf1, zf2, 1e-9, "final state")
        m = 300_000
        wide = np.complex128 if cplx else np.float64
        ref, _ = signal.sosfilt(sos, xd.to_host(0, m).astype(wide), zi=zi.astype(wide))
        assert_close(y1.to_host(0, m), ref, TOL32 if single else 1e-10, "head vs sosfilt(zi)")
        lo = n - m - 60_000
        ref2, zf_ref = signal.sosfilt(sos, xd.to_host(lo, n - lo).astype(wide), zi=np.zeros((nsec, 2), dtype=wide))
        assert_close(y1.to_host(n - m, m), ref2[-m:], TOL32 if single else 1e-10, "tail vs sosfilt")
        assert_close(zf1, zf_ref.ravel(), 1e-9, "final state vs sosfilt")
        # back-to-back launches reuse the look-back slots with a new epoch each: results must not change
        first = y1.to_host(n - w, w)
        for _ in range(5):
            k.filter_dev(xd, y1)
        _ffi.sync()
        with _ffi.option("iir_two_pass", -1):
            k.filter_state_dev(xd, y1, zi=zi_flat, want_zf=False)
        assert np.array_equal(y1.to_host(n - w, w), first)
    finally:
        xd.free()
        y1.free()
        y2.free()


def test_iir_single_pass_not_taken_for_slow_decay():
    """A filter whose transition over one segment (32768 samples) does not vanish keeps the two-pass path (its
    look-back would need a chain): r = 0.99999 resonator, exactness as in the slow-decay test."""
    from scipy import signal
    r, w0 = 0.99999, 0.3
    sos = np.array([[1.0, 0.0, -1.0, 1.0, -2 * r * np.cos(w0), r * r]])
    n = 9_000_000
    x = np.random.default_rng(8).standard_normal(n).astype(np.float64)
    y = mrh.multirate_IIR(sos).filter(x)
    ref = signal.sosfilt(sos, x)
    assert_close(y[-(1 << 20):], ref[-(1 << 20):], 1e-9, "slow decay")


# ----------------------------------------------------------------- float64 overlap-save (fir_ols64.hip)
@pytest.mark.parametrize("dt", [np.complex128, np.float64])
@pytest.mark.parametrize("P", [2, 97, 256, 1024, 2049])
def test_fir_ols64_vs_oracle(dt, P):
    """The float64 overlap-save tile (4096 complex128 points; float64 signals as two real tiles per complex tile):
    .filter and the decimating store against the oracle at the float64 tolerance, ragged lengths, with history, complex
    taps on complex128; and the same calls through the direct float64 kernels."""
    rng = np.random.default_rng(P)
    n = 3 * 4096 + 1234 if P > 1000 else 41_003
    cplx = dt == np.complex128
    x = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dt)
    for ctaps in ((False, True) if cplx else (False,)):
        b = rng.standard_normal(P) / np.sqrt(P) + (1j * rng.standard_normal(P) / np.sqrt(P) if ctaps else 0)
        k = _ffi.FirKernel(b, _ffi.code_of(dt))
        k.set_algo(_ffi.FIR_OLS)
        ref = orc.fir_filter(b, x)
        assert_close(k.filter(x), ref, TOL64, "ols64 filter P=%d" % P)
        assert_close(k.dn(x, 5), ref[::5][:n // 5], TOL64, "ols64 dn P=%d" % P)
        # history in front of the block (n_hist), as the sharded / streaming callers use it
        nh = P - 1
        xd = _ffi.DeviceArray(n - 5000, dt, headroom=max(nh, 1))
        xd.write(x[5000:])
        if nh:
            xd.write(x[5000 - nh:5000], at=-nh)
        yd = _ffi.DeviceArray(n - 5000, dt)
        k.filter_dev(xd, yd, n_hist=nh)
        assert_close(yd.to_host(), ref[5000:], TOL64, "ols64 with history P=%d" % P)
        k2 = _ffi.FirKernel(b, _ffi.code_of(dt))
        k2.set_algo(_ffi.FIR_DIRECT)
        assert_close(k2.filter(x), ref, TOL64, "direct float64 P=%d" % P)


def test_fir_ols64_full_size_windows():
    """2^26 complex128 samples, 1024 taps (the float64 twin of the headline): windows against the oracle, and the
    algorithm AUTO picks is the overlap-save one."""
    import bench
    n = 1 << 26
    b = bench.firwin_lowpass(1024, 0.2)
    k = _ffi.FirKernel(b, _ffi.C128)
    assert k.algo_for(n) == _ffi.FIR_OLS
    xd = _ffi.DeviceArray(n, np.complex128).fill_noise(11)
    yd = _ffi.DeviceArray(n, np.complex128)
    k.filter_dev(xd, yd)
    _ffi.sync()
    for s0 in (0, 3072 * 5000 - 300, n - 9000):
        lo = max(s0 - 1023, 0)
        ref = orc.fir_filter(b, xd.to_host(lo, s0 - lo + 9000))[s0 - lo:]
        assert_close(yd.to_host(s0, 9000), ref, TOL64, "window @%d" % s0)
    xd.free()
    yd.free()


# ----------------------------------------------------------------- round-2 wideners (vectors captured from the reference: G14)
def test_wideners_match_reference():
    """os_filter / oa_filter with the diagnostic matrix, the AM case-study pair around interp24 / deci24, the remaining
    lfilter(b, 1, .) transmitters of digitalcom and the constant-delay Farrow filter, against tests/golden/g14_wideners.npz
    (captured by tests/golden/gen_golden_wideners.py from the real reference; random transmitters under np.random.seed)."""
    from sk_dsp_comm_amd import digitalcom as dcm
    g = load("g14_wideners.npz")
    for name, fn in (("os", ss.os_filter), ("oa", ss.oa_filter)):
        y, ym = fn(g[name + "_x"], g[name + "_h"], 32, mode=1)
        assert_close(y, g[name + "_y"], 1e-11, name + "_filter")
        assert ym.shape == g[name + "_ymat"].shape
        assert_close(ym, g[name + "_ymat"], 1e-12, name + "_filter frame matrix")
    x192, t192, m24 = ss.am_tx(g["am_m"], 0.8, fc=75e3)
    assert np.array_equal(t192, g["am_t192"])
    assert_close(m24, g["am_m24"], 1e-8, "am_tx m24")
    assert_close(x192, g["am_x192"], 1e-8, "am_tx x192")
    m_rx8, t8, m_rx192, x_edet = ss.am_rx(g["am_x192"])
    assert np.array_equal(x_edet, g["am_edet"]) and np.array_equal(t8, g["am_t8"])
    assert_close(m_rx8, g["am_rx8"], 1e-8, "am_rx 8 ksps")
    assert_close(m_rx192, g["am_rx192"], 1e-8, "am_rx 192 ksps")
    np.random.seed(2024)
    x, b, d = dcm.qam_bb(300, 8, '16qam', 'src', 0.25)
    assert np.array_equal(d, g["qam_d"]) and np.allclose(b, g["qam_b"], rtol=0, atol=1e-15)
    assert_close(x, g["qam_x"], 1e-11, "qam_bb")
    np.random.seed(2025)
    x, b, d = dcm.qam_bb(200, 4, 'qpsk', 'rect')
    assert np.array_equal(d, g["qpsk_d"])
    assert_close(x, g["qpsk_x"], 1e-11, "qam_bb qpsk")
    np.random.seed(2026)
    x, b, d = dcm.mpsk_bb(256, 10, 8, 'rc', 0.35, 5)
    assert np.array_equal(d, g["mpsk_d"]) and np.allclose(b, g["mpsk_b"], rtol=0, atol=1e-15)
    assert_close(x, g["mpsk_x"], 1e-11, "mpsk_bb")
    np.random.seed(2027)
    x, b, d = dcm.mpsk_bb(128, 6, 4, 'rect')
    assert_close(x, g["mpsk4_x"], 1e-11, "mpsk_bb qpsk rotation")
    np.random.seed(2028)
    x, b, d = dcm.rz_bits(500, 12, 'src', 0.5, 4)
    assert np.array_equal(d, g["rz_d"])
    assert_close(x, g["rz_x"], 1e-11, "rz_bits")
    np.random.seed(2029)
    y, d = dcm.gmsk_bb(400, 8, 1, 0.3)
    assert np.array_equal(d, g["gmsk_d"])
    assert_close(y, g["gmsk_y"], 1e-9, "gmsk_bb")
    np.random.seed(2030)
    y, d = dcm.gmsk_bb(300, 6, 0)
    assert_close(y, g["msk_y"], 1e-9, "msk")
    assert_close(dcm.time_delay(g["td_x"], 1.37, 4), g["td_y"], 1e-12, "time_delay")
    assert_close(dcm.time_delay(g["td_x"], 2.0, 6), g["td_y2"], 1e-12, "time_delay integer")
    # a delay per sample: the reference's time-varying Farrow loop (digitalcom.py:1132-1160)
    assert_close(dcm.time_delay(g["td_x"], g["td_d"], 4), g["td_y3"], 1e-13, "time_delay d[k]")
    assert_close(dcm.time_delay(g["td_x"], g["td_d6"], 6), g["td_y4"], 1e-13, "time_delay d[k], n = 6")
    with pytest.raises(ValueError):
        dcm.time_delay(g["td_x"], np.full(len(g["td_x"]), 3.5), 4)


@pytest.mark.parametrize("dt", [np.float32, np.complex64])
def test_iir_single_pass_soak(dt):
    """400 back-to-back single-pass launches alternating two lengths (different segment counts, look-back slots reused
    with a new epoch every launch, the ticket dispensers running on): every output equals that of the first launch of
    its length."""
    from scipy import signal
    sos = signal.cheby1(6, 0.05, 0.2, output="sos")
    k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
    per = 128 if dt == np.float32 else 64
    sizes = (300 * 256 * per + 777, 517 * 256 * per - 3)
    xd = _ffi.DeviceArray(max(sizes), dt).fill_noise(5)
    yd = _ffi.DeviceArray(max(sizes), dt)
    ref = {}
    w = 1 << 18
    with _ffi.option("iir_two_pass", -1):
        for it in range(400):
            n = sizes[it & 1] if it % 7 else sizes[0]
            k.filter_dev(xd, yd, n)
            if it < 2 or it % 50 == 49:
                got = (yd.to_host(n - w, w), yd.to_host(n // 2, 4096))
                if n not in ref:
                    ref[n] = got
                else:
                    assert np.array_equal(got[0], ref[n][0]) and np.array_equal(got[1], ref[n][1]), it
    m = 100_000
    want = signal.sosfilt(sos, xd.to_host(0, m).astype(np.complex128 if dt == np.complex64 else np.float64))
    k.filter_dev(xd, yd, sizes[0])
    assert_close(yd.to_host(0, m), want, TOL32, "after the soak")
    xd.free()
    yd.free()
