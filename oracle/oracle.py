"""CPU oracle for the streaming-filter hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (scikit-dsp-comm_amd/sk_dsp_comm_amd) never
does, and it raises if its HIP library is missing instead of falling back here.

Every function restates what the reference executes for this path, in float64 /
complex128 exactly as the reference computes it (scipy promotes to double
because `a=[1]` is an int64 array -- SURVEY.md section 3.1), and cites the reference
line it follows.  Heavy loops live in oracle.c (same directory, built by
`make -C oracle`); pure-NumPy fallbacks exist for small cases.

Parity of this oracle is PINNED: tests/test_oracle_golden.py checks every
function against tests/golden/*.npz, captured from the real reference by
tests/golden/gen_golden.py (exact for upsample/downsample/cic and the DF2T IIR
paths, <=1e-12 relative for the FIR dot products whose BLAS summation order is
not reproducible).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle.c -> liboracle.so (gcc).  Building the checker is not using it."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        fp = ctypes.POINTER(ctypes.c_float)
        i64 = ctypes.c_int64
        for name in ("orc_fir_rr", "orc_fir_rc", "orc_fir_cc"):
            f = getattr(L, name)
            f.argtypes = [dp, ctypes.c_int, dp, i64, dp, i64, dp, ctypes.c_int]
            f.restype = None
        for name in ("orc_fir_rc_f32in", "orc_fir_rr_f32in"):
            f = getattr(L, name)
            f.argtypes = [dp, ctypes.c_int, fp, i64, dp, ctypes.c_int]
            f.restype = None
        L.orc_sosfilt.argtypes = [dp, ctypes.c_int, dp, i64, ctypes.c_int, dp, dp]
        L.orc_sosfilt.restype = ctypes.c_int
        L.orc_sosfilt_f32in.argtypes = [dp, ctypes.c_int, fp, i64, dp]
        L.orc_sosfilt_f32in.restype = ctypes.c_int
        L.orc_lfilter.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, i64, ctypes.c_int, dp]
        L.orc_lfilter.restype = ctypes.c_int
        L.orc_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _as_c128_or_f64(x):
    """The reference's promotion: anything real -> float64, anything complex -> complex128."""
    x = np.asarray(x)
    if np.iscomplexobj(x):
        return np.ascontiguousarray(x, dtype=np.complex128), True
    return np.ascontiguousarray(x, dtype=np.float64), False


# ----------------------------------------------------------------------------
# sigsys primitives
# ----------------------------------------------------------------------------
def upsample(x, L):
    """sigsys.upsample (sigsys.py:3031-3053): y[n*L] = x[n], zeros elsewhere.

    The factor is int(L-1)+1 and the result dtype is result_type(x, float64)
    because the reference hstacks x with a float64 zeros matrix."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("oracle.upsample: 1-D input only")
    n = len(x)
    Lz = int(L - 1)
    if Lz < 0:
        raise ValueError("negative dimensions are not allowed")
    Li = Lz + 1
    y = np.zeros(n * Li, dtype=np.result_type(x.dtype, np.float64))
    y[::Li] = x
    return y


def downsample(x, M, p=0):
    """sigsys.downsample (sigsys.py:3056-3083): y[k] = x[k*M+p], k < floor(N/M)."""
    if not isinstance(M, int):
        raise TypeError("M must be an int")
    x = np.asarray(x)
    nk = len(x) // M
    if not (-M <= p < M):
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (p, M))
    pp = p % M
    return np.ascontiguousarray(x[: nk * M].reshape(nk, M)[:, pp])


def cic(m, k):
    """sigsys.cic (sigsys.py:62-93): k cascaded length-m boxcars, unit DC gain."""
    if k == 1:
        b = np.ones(m)
    else:
        h = np.ones(m)
        b = h
        for _ in range(1, k):
            b = np.convolve(b, h)
    return b / np.sum(b)


# ----------------------------------------------------------------------------
# FIR (multirate_FIR.filter/up/dn -- multirate_helper.py:104-127)
# ----------------------------------------------------------------------------
def fir_filter(b, x, hist=None, nthreads=1):
    """lfilter(b,[1],x) == np.convolve(b,x)[:len(x)] in float64/complex128.

    `hist` (optional) = the samples preceding x[0] (last element is x[-1]); the
    reference always starts from zero state, hist is only used by the sharding
    tests to restate the halo semantics."""
    b = np.asarray(b)
    x64, xc = _as_c128_or_f64(x)
    if x64.ndim != 1:
        return np.stack([fir_filter(b, row, None, nthreads) for row in x64.reshape(-1, x64.shape[-1])]).reshape(
            x64.shape[:-1] + (x64.shape[-1],))
    n = len(x64)
    if n == 0:
        raise ValueError("v cannot be empty")
    bc = np.iscomplexobj(b)
    L = _lib()
    h64 = None
    nh = 0
    if hist is not None and len(hist):
        h64 = np.ascontiguousarray(hist, dtype=np.complex128 if (xc or bc) else np.float64)
        nh = len(h64)
    if bc:
        bb = np.ascontiguousarray(b, dtype=np.complex128)
        xx = x64.astype(np.complex128)
        y = np.empty(n, dtype=np.complex128)
        L.orc_fir_cc(_dp(bb.view(np.float64)), len(bb), _dp(xx.view(np.float64)), n,
                     _dp(h64.view(np.float64)) if h64 is not None else None, nh, _dp(y.view(np.float64)), nthreads)
        return y
    bb = np.ascontiguousarray(b, dtype=np.float64)
    if xc:
        y = np.empty(n, dtype=np.complex128)
        L.orc_fir_rc(_dp(bb), len(bb), _dp(x64.view(np.float64)), n,
                     _dp(h64.view(np.float64)) if h64 is not None else None, nh, _dp(y.view(np.float64)), nthreads)
    else:
        y = np.empty(n, dtype=np.float64)
        L.orc_fir_rr(_dp(bb), len(bb), _dp(x64), n, _dp(h64) if h64 is not None else None, nh, _dp(y), nthreads)
    return y


def fir_up(b, x, L_change=12):
    """multirate_FIR.up (multirate_helper.py:112-118): lfilter(b,[1], L*upsample(x,L))."""
    return fir_filter(b, L_change * upsample(x, L_change))


def fir_dn(b, x, M_change=12):
    """multirate_FIR.dn (multirate_helper.py:121-127): downsample(lfilter(b,[1],x), M)."""
    return downsample(fir_filter(b, x), M_change)


def fir_filter_f32in_timed(b, x32, nthreads=1):
    """cpu_baseline leg: c64/f32 input, float64 accumulation, complex128/float64 out
    (the shape of the reference call on the benchmark configs)."""
    L = _lib()
    bb = np.ascontiguousarray(b, dtype=np.float64)
    x32 = np.ascontiguousarray(x32)
    n = len(x32)
    if x32.dtype == np.complex64:
        y = np.empty(n, dtype=np.complex128)
        L.orc_fir_rc_f32in(_dp(bb), len(bb), _fp(x32.view(np.float32)), n, _dp(y.view(np.float64)), nthreads)
    elif x32.dtype == np.float32:
        y = np.empty(n, dtype=np.float64)
        L.orc_fir_rr_f32in(_dp(bb), len(bb), _fp(x32), n, _dp(y), nthreads)
    else:
        raise TypeError("float32/complex64 only")
    return y


# ----------------------------------------------------------------------------
# IIR: sosfilt (multirate_IIR -- multirate_helper.py:159-192)
# ----------------------------------------------------------------------------
def sos_filter(sos, x):
    """scipy.signal.sosfilt(sos, x) restated (DF2T per biquad, zero state)."""
    sos = np.atleast_2d(np.asarray(sos))
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError("sos array must be shape (n_sections, 6)")
    if not (sos[:, 3] == 1).all():
        raise ValueError("sos[:, 3] should be all ones")
    x64, xc = _as_c128_or_f64(x)
    if x64.ndim != 1:
        return np.stack([sos_filter(sos, row) for row in x64.reshape(-1, x64.shape[-1])]).reshape(x64.shape)
    n = len(x64)
    if n == 0:
        raise ValueError("cannot reshape array of size 0 into shape (0)")
    s64 = np.ascontiguousarray(sos, dtype=np.float64)
    y = np.empty_like(x64)
    rc = _lib().orc_sosfilt(_dp(s64), s64.shape[0], _dp(x64.view(np.float64)), n, 2 if xc else 1, None,
                            _dp(y.view(np.float64)))
    if rc != 0:
        raise ValueError("orc_sosfilt failed: %d" % rc)
    return y


def sos_filter_py(sos, x):
    """Pure-Python DF2T loop (small cases only) -- cross-checks the C loop."""
    sos = np.asarray(sos, dtype=np.float64)
    x = np.asarray(x)
    y = np.zeros(len(x), dtype=np.result_type(x.dtype, np.float64))
    z = np.zeros((sos.shape[0], 2), dtype=y.dtype)
    for i in range(len(x)):
        xc = x[i]
        for s in range(sos.shape[0]):
            xn = xc
            xc = sos[s, 0] * xn + z[s, 0]
            z[s, 0] = sos[s, 1] * xn - sos[s, 4] * xc + z[s, 1]
            z[s, 1] = sos[s, 2] * xn - sos[s, 5] * xc
        y[i] = xc
    return y


def sos_up(sos, x, L_change=12):
    """multirate_IIR.up (multirate_helper.py:177-183)."""
    return sos_filter(sos, L_change * upsample(x, L_change))


def sos_dn(sos, x, M_change=12):
    """multirate_IIR.dn (multirate_helper.py:186-192)."""
    return downsample(sos_filter(sos, x), M_change)


def sos_filter_f32in_timed(sos, x32):
    s64 = np.ascontiguousarray(sos, dtype=np.float64)
    x32 = np.ascontiguousarray(x32, dtype=np.float32)
    y = np.empty(len(x32), dtype=np.float64)
    rc = _lib().orc_sosfilt_f32in(_dp(s64), s64.shape[0], _fp(x32), len(x32), _dp(y))
    if rc != 0:
        raise ValueError("orc_sosfilt_f32in failed: %d" % rc)
    return y


# ----------------------------------------------------------------------------
# IIR: lfilter(b,a,x) (rate_change -- multirate_helper.py:69-83; interp24/deci24)
# ----------------------------------------------------------------------------
def lfilter(b, a, x):
    """scipy.signal.lfilter(b,a,x): DF2T, a[0]-normalised, zero initial state."""
    b = np.ascontiguousarray(np.atleast_1d(b), dtype=np.float64)
    a = np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
    if len(a) == 1:
        return fir_filter(b / a[0], x)
    x64, xc = _as_c128_or_f64(x)
    n = len(x64)
    y = np.empty_like(x64)
    rc = _lib().orc_lfilter(_dp(b), len(b), _dp(a), len(a), _dp(x64.view(np.float64)), n, 2 if xc else 1,
                            _dp(y.view(np.float64)))
    if rc != 0:
        raise ValueError("orc_lfilter failed: %d" % rc)
    return y


def rate_change_up(b, a, M, x):
    """rate_change.up (multirate_helper.py:69-75): lfilter(b,a, M*upsample(x,M))."""
    return lfilter(b, a, M * upsample(x, M))


def rate_change_dn(b, a, M, x):
    """rate_change.dn (multirate_helper.py:77-83): downsample(lfilter(b,a,x), M)."""
    return downsample(lfilter(b, a, x), M)


def max_threads():
    return int(_lib().orc_max_threads())
