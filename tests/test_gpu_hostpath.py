"""Host-array (NumPy in / NumPy out) path on long vectors: chunk pipeline and multi-slot dealing (run with -m gpu).

The reference call is one function call on one array (multirate_helper.py:104-127, 169-192).  Long arrays are cut into
chunks that are exact continuations of each other and pipelined over PCIe; with several slots bound the chunks of a FIR are
dealt to all of them.  Everything here must equal the single-shot path and the oracle; the one-GPU box binds two slots
to the same GPU, which runs every line of the multi-GPU code except the second PCIe link."""
import os
import subprocess
import sys

import numpy as np
import pytest

from sk_dsp_comm_amd import _ffi, multirate_helper as mrh, config
from oracle import oracle as orc
from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    _ffi.init()
    yield


def cnoise(rng, n):
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).astype(np.complex64)


@pytest.mark.parametrize("dt", [np.complex64, np.float32, np.float64])
def test_pipelined_fir_equals_single_shot(dt):
    """filter / up / dn / updn of a vector that spans many chunks (host_chunk_log2 = 16) against the same calls with the
    pipeline off, and windows against the oracle: chunk boundaries are invisible."""
    rng = np.random.default_rng(3)
    n = 1_000_003
    x = cnoise(rng, n) if dt == np.complex64 else rng.standard_normal(n).astype(dt)
    b = rng.standard_normal(301) / 17
    k = _ffi.FirKernel(b, _ffi.code_of(dt))
    with _ffi.option("host_chunk_log2", 16):
        got = [k.filter(x), k.up(x[:200_000], 3), k.dn(x, 5), k.updn(x[:300_001], 4, 3), k.filter(x, wide=True)]
    with _ffi.option("host_pipeline", 0):
        want = [k.filter(x), k.up(x[:200_000], 3), k.dn(x, 5), k.updn(x[:300_001], 4, 3), k.filter(x, wide=True)]
    for g, w, name in zip(got, want, ("filter", "up", "dn", "updn", "filter wide")):
        assert g.dtype == w.dtype and g.shape == w.shape
        e_max, e_l2 = rel_err(g, w)
        assert e_max <= (2e-6 if dt != np.float64 else 1e-12), (name, e_max)   # (algorithm choice may differ per chunk length)
    s0 = 65536 * 7 - 500
    ref = orc.fir_filter(b, x[s0 - 300:s0 + 3000])[300:]
    assert max(rel_err(got[0][s0:s0 + 3000], ref)) <= (1e-6 if dt != np.float64 else 1e-11)
    ref = orc.fir_dn(b, x[:400_000], 5)
    assert max(rel_err(got[2][:80_000], ref)) <= (1e-6 if dt != np.float64 else 1e-11)


@pytest.mark.parametrize("dt", [np.float32, np.complex64, np.float64])
def test_pipelined_iir_carries_state_across_chunks(dt):
    from scipy import signal
    import bench
    sos = bench.elliptic_bpf_sos()
    rng = np.random.default_rng(4)
    n = 700_001
    x = cnoise(rng, n) if dt == np.complex64 else rng.standard_normal(n).astype(dt)
    f = mrh.multirate_IIR(sos)
    with _ffi.option("host_chunk_log2", 16):
        y = f.filter(x)
    ref = signal.sosfilt(sos, x.astype(np.complex128 if dt == np.complex64 else np.float64))
    assert max(rel_err(y, ref)) <= (1e-6 if dt != np.float64 else 1e-10)


def test_two_slots_on_one_gpu_deal_the_chunks():
    """skdsp_init_devices([0, 0]): two slots (streams, workspaces, handle clones, worker threads) on the one GPU of this
    box.  The multi-slot FIR host call must equal the oracle; reference surface (multirate_FIR) with reference dtypes."""
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'scikit-dsp-comm_amd'))
from sk_dsp_comm_amd import _ffi, multirate_helper as mrh
from oracle import oracle as orc
assert _ffi.init_devices([0, 0, 0]) == 3
_ffi.set_option('host_chunk_log2', 17)
rng = np.random.default_rng(9)
n = 3_000_001
x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).astype(np.complex64)
b = np.hamming(1024) * np.sinc(0.2 * (np.arange(1024) - 511.5)) * 0.2
f = mrh.multirate_FIR(b)
for rep in range(3):
    y = f.filter(x)
assert y.dtype == np.complex128
for s0 in (0, 131072 * 7 - 2000, 131072 * 11 + 5, n - 5000):
    lo = max(s0 - 1023, 0)
    ref = orc.fir_filter(b, x[lo:s0 + 5000])[s0 - lo:]
    e = np.max(np.abs(y[s0:s0 + 5000] - ref)) / np.max(np.abs(ref))
    assert e < 1e-6, (s0, e)
_ffi.set_option('host_multi_slot', 0)
y1 = f.filter(x)
assert np.max(np.abs(y1 - y)) / np.max(np.abs(y)) < 1e-6
yd = f.dn(x, 12); yu = f.up(x[:200000], 12)
_ffi.set_option('host_multi_slot', 1)
assert np.array_equal(f.dn(x, 12), yd) and np.array_equal(f.up(x[:200000], 12), yu)
# skdsp_fir_filter_sharded: the caller chooses how many of the bound slots take part (0 = all); asking for more is an argument error
k = _ffi.FirKernel(b, _ffi.C64)
y0 = k.filter(x)
for ng in (0, 1, 2, 3):
    ys = k.filter_sharded(x, ng)
    assert np.max(np.abs(ys - y0)) / np.max(np.abs(y0)) < 1e-6, ng
try:
    k.filter_sharded(x, 4)
    raise SystemExit('ngpu above the bound slots was accepted')
except ValueError:
    pass
print('SLOTS_OK')
""" % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert b"SLOTS_OK" in out.stdout, out.stdout.decode()[-3000:]


@pytest.mark.parametrize("dt,P", [(np.float32, 10_001), (np.complex64, 9_000), (np.float64, 7_001), (np.complex128, 5_000)])
def test_fir_longer_than_one_launch_is_partitioned(dt, P):
    """The reference takes any tap count (lfilter).  Beyond what one launch holds (4097 taps overlap-save, ~3000 in the
    float64 direct kernels) the taps are cut into segments applied to delayed inputs and summed: .filter, .dn and block
    streaming with history, against the oracle."""
    rng = np.random.default_rng(P)
    n = 60_000
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if np.dtype(dt).kind == "c" else 0)
    x = x.astype(dt)
    b = rng.standard_normal(P) / np.sqrt(P)
    tol = 2e-6 if np.dtype(dt).itemsize <= 8 and dt != np.float64 else 1e-11
    f = mrh.multirate_FIR(b)
    ref = orc.fir_filter(b, x)
    assert max(rel_err(f.filter(x), ref)) <= tol
    assert max(rel_err(f.dn(x, 3), ref[::3][:n // 3])) <= tol
    y1, z = f.filter_stream(x[:25_000])
    y2, _ = f.filter_stream(x[25_000:], zi=z)
    assert max(rel_err(np.concatenate([y1, y2]), ref)) <= tol


def test_precision_switch():
    """config.precision: 'single' runs float64 callers through the float32 engines (1e-6), 'double' runs float32 callers in
    float64 (the reference's own arithmetic: 1e-12, also in the stop band)."""
    rng = np.random.default_rng(2)
    b = np.ones(1024) / 1024
    x64 = np.cos(2 * np.pi * 0.0123 * np.arange(50_000))       # attenuated by 32 dB: stop band of the boxcar
    ref = orc.fir_filter(b, x64)
    f = mrh.multirate_FIR(b)
    old = config.precision
    try:
        config.precision = "single"
        y = f.filter(x64)
        assert y.dtype == np.float64 and 1e-9 < float(np.max(np.abs(y - ref))) <= 1e-6   # float32 arithmetic, input-relative bound
        config.precision = "double"
        y = f.filter(x64.astype(np.float32))
        ref32 = orc.fir_filter(b, x64.astype(np.float32))
        assert max(rel_err(y, ref32)) <= 1e-11
        config.precision = "nonsense"
        with pytest.raises(ValueError):
            f.filter(x64)
    finally:
        config.precision = old


def test_threads_share_objects_and_library():
    """Four caller threads, one shared multirate_FIR and one shared multirate_IIR, mixed result widths: calls take turns on
    slot 0 (per-slot lock), the width switch and its call are one critical section, results equal the single-threaded ones."""
    import threading
    from scipy import signal
    rng = np.random.default_rng(12)
    b = rng.standard_normal(200) / 14
    sos = signal.butter(6, 0.2, output="sos")
    f, g = mrh.multirate_FIR(b), mrh.multirate_IIR(sos)
    xs = [cnoise(rng, 200_000 + 1000 * i) for i in range(4)]
    want = [(orc.fir_filter(b, x), signal.sosfilt(sos, x.astype(np.complex128))) for x in xs]
    errs = []

    def work(i):
        try:
            for rep in range(6):
                old = config.strict_dtype
                y = f.filter(xs[i])
                z = g.filter(xs[i])
                k = _ffi.FirKernel(b, _ffi.C64)          # and a private low-level object with explicit widths
                y2 = k.filter(xs[i], wide=bool((i + rep) & 1))
                assert y2.dtype == (np.complex128 if (i + rep) & 1 else np.complex64)
                for got, ref in ((y, want[i][0]), (z, want[i][1]), (y2, want[i][0])):
                    e = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
                    assert e < 1e-6, e
        except Exception as ex:  # noqa: BLE001
            errs.append((i, repr(ex)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("dt,P,L,M", [(np.float32, 50_000, 12, 1), (np.complex64, 17_000, 4, 3), (np.float64, 26_000, 12, 1),
                                       (np.complex128, 9_000, 4, 3)])
def test_interpolators_longer_than_one_launch(dt, P, L, M):
    """.up / the fused L-over-M resampler with more taps per phase than one launch takes (4097 / 2049): tap segments whose
    length is a multiple of lcm(L, M), each applied to the input shortened by its delay and summed."""
    rng = np.random.default_rng(P + L)
    n = 3000 - 3000 % M
    x = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if np.dtype(dt).kind == "c" else 0)).astype(dt)
    b = rng.standard_normal(P) / np.sqrt(P)
    tol = 2e-6 if dt in (np.float32, np.complex64) else 1e-11
    f = mrh.multirate_FIR(b)
    ref = orc.fir_up(b, x, L)
    if M == 1:
        assert max(rel_err(f.up(x, L), ref)) <= tol
    else:
        assert max(rel_err(f.updn(x, L, M), ref[::M][:(n * L) // M])) <= tol
