"""Rate-change and CIC-tap primitives with the signatures of sk_dsp_comm.sigsys.

  upsample(x, L)       /root/reference/src/sk_dsp_comm/sigsys.py:3031-3053
  downsample(x, M, p)  /root/reference/src/sk_dsp_comm/sigsys.py:3056-3083
  cic(m, k)            /root/reference/src/sk_dsp_comm/sigsys.py:62-93

  os_filter(x,h,N,mode) /root/reference/src/sk_dsp_comm/sigsys.py:482-540   ("next" row)
  oa_filter(x,h,N,mode) /root/reference/src/sk_dsp_comm/sigsys.py:543-598

upsample/downsample run on the GPU (resample.hip) and are bit-exact index moves; cic
is host-side coefficient generation (a few dozen float64 taps) and stays in NumPy.
Error conventions follow the reference (tests/golden/g10_conventions.json).
"""
from logging import getLogger

import numpy as np

from . import _ffi
from . import config

log = getLogger(__name__)


def _gpu_dtype(x):
    """Map an input array to one of the four device dtypes (value-preserving)."""
    dt = x.dtype
    if dt in (np.float32, np.complex64, np.float64, np.complex128):
        return x
    if np.issubdtype(dt, np.complexfloating):
        return x.astype(np.complex128)
    if dt == np.float16:
        return x.astype(np.float32)
    return x.astype(np.float64)  # integers / bool: what the reference's promotion gives


def cic(m, k):
    """FIR taps of k cascaded length-m boxcars with unit DC gain (sigsys.py:62-93)."""
    if k == 1:
        b = np.ones(m)
    else:
        h = np.ones(m)
        b = h
        for _ in range(1, k):
            b = np.convolve(b, h)  # cascade by convolving impulse responses
    return b / np.sum(b)


def upsample(x, L):
    """Insert L-1 zeros between samples: y[n*L] = x[n] (sigsys.py:3050-3053).

    Like the reference the stuffing factor is int(L-1)+1 and the result is
    float64/complex128 (the reference hstacks with a float64 zeros matrix)."""
    if not hasattr(x, "reshape"):
        raise AttributeError("'%s' object has no attribute 'reshape'" % type(x).__name__)
    n_in = len(x)
    if x.ndim != 1:
        raise ValueError("cannot reshape array of size %d into shape (%d,1)" % (x.size, n_in))
    Lz = int(L - 1)
    if Lz < 0:
        raise ValueError("negative dimensions are not allowed")
    Li = Lz + 1
    out_dt = np.result_type(x.dtype, np.float64)
    if n_in == 0:
        return np.zeros(0, dtype=out_dt)
    xg = np.ascontiguousarray(_gpu_dtype(x))
    y = _ffi.upsample(xg, Li)
    return y.astype(out_dt, copy=False) if config.strict_dtype else y


def downsample(x, M, p=0):
    """Keep every M-th sample starting at phase p: y[k] = x[k*M+p] (sigsys.py:3078-3083).

    Returns a fresh contiguous array (the reference returns a strided view of x; the
    values are identical)."""
    if not isinstance(M, int):
        raise TypeError("M must be an int")
    if not hasattr(x, "reshape"):
        raise AttributeError("'%s' object has no attribute 'reshape'" % type(x).__name__)
    nk = int(np.floor(len(x) / M))  # ZeroDivisionError for M == 0, like the reference
    if x.ndim != 1:
        raise ValueError("cannot reshape array of size %d into shape (%d,%d)" % (x.size, nk, M))
    if not (-M <= p < M):
        raise IndexError("index %d is out of bounds for axis 1 with size %d" % (p, M))
    p = int(p) % M
    if nk == 0:
        return np.zeros(0, dtype=x.dtype)
    src = np.ascontiguousarray(x)
    # a pure element move: run it on the bit pattern (4/8/16-byte elements map to f32/f64/c128)
    isz = src.dtype.itemsize
    if isz == 4:
        y = _ffi.downsample(src.view(np.float32), M, p).view(src.dtype)
    elif isz == 8:
        y = _ffi.downsample(src.view(np.float64), M, p).view(src.dtype)
    elif isz == 16:
        y = _ffi.downsample(src.view(np.complex128), M, p).view(src.dtype)
    else:  # 1- and 2-byte elements: widen exactly, move, narrow
        wide = src.astype(np.float32 if src.dtype == np.float16 else np.int32)
        y = _ffi.downsample(wide.view(np.float32), M, p).view(wide.dtype).astype(src.dtype)
    return y


def _transform_domain_fir(x, h, N, mode, name):
    """Shared body of os_filter / oa_filter: both return real(lfilter(h, 1, x)) as float64
    (the reference takes np.real of every inverse FFT frame).  The frame size N only
    parameterises the reference's Python frame loop; here the filtering runs in the GPU
    overlap-save engine (fir_ols.hip) with its own 8192-point tiles."""
    from . import multirate_helper as mrh
    P = len(h)
    L = int(N) - P + 1
    if L <= 0:
        raise ValueError("%s: FFT size N=%d must exceed the filter length P=%d - 1" % (name, int(N), P))
    if mode == 1:
        raise NotImplementedError("%s(mode=1): the per-frame diagnostic matrix is a teaching aid of the "
                                  "reference's Python loop and is not produced by the GPU path" % name)
    x = np.asarray(x)
    if len(x) == 0:
        return np.zeros(0)
    saved = config.strict_dtype
    config.strict_dtype = True
    try:
        y = mrh.multirate_FIR(np.asarray(h)).filter(x)
    finally:
        config.strict_dtype = saved
    return np.ascontiguousarray(np.real(y), dtype=np.float64)


def os_filter(x, h, N, mode=0):
    """Overlap-and-save FIR filtering (sigsys.py:482-540): y = real(lfilter(h, 1, x))."""
    return _transform_domain_fir(x, h, N, mode, "os_filter")


def oa_filter(x, h, N, mode=0):
    """Overlap-and-add FIR filtering (sigsys.py:543-598): y = real(lfilter(h, 1, x))."""
    return _transform_domain_fir(x, h, N, mode, "oa_filter")
