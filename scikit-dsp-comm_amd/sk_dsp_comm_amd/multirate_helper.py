"""Multirate filter objects with the interface of sk_dsp_comm.multirate_helper.

  rate_change    /root/reference/src/sk_dsp_comm/multirate_helper.py:45-83
  multirate_FIR  /root/reference/src/sk_dsp_comm/multirate_helper.py:85-143
  multirate_IIR  /root/reference/src/sk_dsp_comm/multirate_helper.py:146-208

Same constructors, attributes, defaults, log lines and output lengths; the arithmetic
that the reference hands to scipy.signal.lfilter / sosfilt runs in HIP kernels on an
MI355X (direct/polyphase and FFT overlap-save FIR, exact affine-scan IIR).  Every call
starts from zero filter state, like the reference (it never passes zi).

dtype policy: float32/complex64 inputs are filtered in float32 on the
GPU (float64 state inside the IIR), float64/complex128 inputs in float64; results are
returned as float64/complex128 exactly like the reference unless
`sk_dsp_comm_amd.config.strict_dtype = False`.
"""
import warnings
from logging import getLogger

import numpy as np

from . import _ffi
from . import config
from . import sigsys as ssd

log = getLogger(__name__)

_MAX_SOS = 4096  # sections per device cascade (the library runs more than 8 as consecutive groups on the device); beyond: chained cascades


def _signal(x, coef_complex=False):
    """-> (contiguous array in a device dtype, reference result dtype)."""
    x = np.asarray(x)
    cplx = np.iscomplexobj(x) or coef_complex
    ref_dt = np.complex128 if cplx else np.float64
    mode = config.precision
    if mode not in ("input", "double", "single"):
        raise ValueError("config.precision must be 'input', 'double' or 'single' (got %r)" % (mode,))
    narrow = x.dtype in (np.float32, np.complex64, np.float16)
    if mode == "double" or (mode == "input" and not narrow):
        dev_dt = np.complex128 if cplx else np.float64     # double (and integer) callers: float64 kernels
    else:
        dev_dt = np.complex64 if cplx else np.float32
    return np.ascontiguousarray(x, dtype=dev_dt), ref_dt


def _finish(y, ref_dt):
    return y.astype(ref_dt, copy=False) if config.strict_dtype else y


def _wide():
    """strict_dtype: float32/complex64 kernels hand back float64/complex128 (widened on the device,
    which is 6x cheaper than astype() on the host); _finish() is then a no-op."""
    return bool(config.strict_dtype)


def _fingerprint(*arrays):
    """Cheap identity of the coefficient arrays an object currently holds (the reference reads its public
    attributes on every call, so obj.b = new_taps must take effect on the next call)."""
    out = []
    for a in arrays:
        a = np.asarray(a)
        out.append((a.dtype.str, a.shape, hash(a.tobytes())))
    return tuple(out)


class _KernelCache:
    """One device handle per signal dtype, created lazily and dropped when the coefficients change."""

    def __init__(self, make, coeffs):
        self._make = make
        self._coeffs = coeffs     # () -> tuple of the arrays the handles were built from
        self._k = {}
        self._fp = None

    def get(self, np_dtype):
        fp = _fingerprint(*self._coeffs())
        if fp != self._fp:
            self._k = {}
            self._fp = fp
        code = _ffi.code_of(np_dtype)
        k = self._k.get(code)
        if k is None:
            k = self._k[code] = self._make(code)
        return k


class multirate_FIR(object):
    """FIR filter / FIR interpolator / FIR decimator (multirate_helper.py:85-143)."""

    def __init__(self, b):
        self.N_forder = len(b)
        self.b = b
        log.info('FIR filter taps = %d' % self.N_forder)
        self._kern = _KernelCache(lambda code: _ffi.FirKernel(np.asarray(self.b), code), lambda: (self.b,))

    @property
    def _bc(self):
        return bool(np.iscomplexobj(np.asarray(self.b)))

    # --- reference surface --------------------------------------------------
    def filter(self, x):
        """y = lfilter(b, [1], x)  (multirate_helper.py:104-109)"""
        return self._filter(x, bool(config.strict_dtype))

    def _filter(self, x, strict):
        """filter() with the result-dtype policy as an argument (callers inside the package that need the reference's
        dtypes whatever the global switch says pass strict=True instead of flipping config.strict_dtype around the call:
        that was not thread-safe)."""
        xg, ref_dt = _signal(x, self._bc)
        if xg.size == 0:
            raise ValueError("v cannot be empty")
        k = self._kern.get(xg.dtype)
        if xg.ndim > 1:   # N-D: every row of the last axis in one call (rows laid end to end behind Ntaps-1 zeros)
            x2 = xg.reshape(-1, xg.shape[-1])
            try:
                y = k.filter_rows(x2, wide=strict).reshape(xg.shape)
            except NotImplementedError:   # the rows do not fit one staged block: row by row, each through the chunk pipeline
                y = np.stack([k.filter(np.ascontiguousarray(r), wide=strict) for r in x2]).reshape(xg.shape)
        else:
            y = k.filter(xg, wide=strict)
        return y.astype(ref_dt, copy=False) if strict else y

    def up(self, x, L_change=12):
        """y = lfilter(b, [1], L*upsample(x, L))  (multirate_helper.py:112-118), polyphase on the GPU."""
        Li, gain_fix = _stuff_factor(x, L_change)
        xg, ref_dt = _signal(x, self._bc)
        if xg.size == 0:
            raise ValueError("v cannot be empty")
        k = self._kern.get(xg.dtype)
        y = k.up(xg, Li, wide=_wide())
        if gain_fix != 1.0:
            y = y * y.dtype.type(gain_fix) if not np.iscomplexobj(y) else y * gain_fix
        return _finish(y, ref_dt)

    def dn(self, x, M_change=12):
        """y = downsample(lfilter(b, [1], x), M)  (multirate_helper.py:121-127); only kept outputs are computed."""
        if not isinstance(M_change, int):
            raise TypeError("M must be an int")
        xg, ref_dt = _signal(x, self._bc)
        if xg.size == 0:
            raise ValueError("v cannot be empty")
        if xg.ndim != 1:
            raise ValueError("cannot reshape array of size %d into shape (%d,%d)"
                             % (xg.size, int(np.floor(len(xg) / M_change)), M_change))
        nk = int(np.floor(len(xg) / M_change))  # ZeroDivisionError for 0, like downsample()
        if nk == 0:
            return np.zeros(0, dtype=ref_dt if config.strict_dtype else xg.dtype)
        k = self._kern.get(xg.dtype)
        return _finish(k.dn(xg, M_change, wide=_wide()), ref_dt)

    # --- extension: fused rational resampler (BASELINE.json config 3) --------
    def updn(self, x, L_change, M_change):
        """== sigsys.downsample(self.up(x, L_change), M_change), computed in one kernel."""
        if not isinstance(M_change, int):
            raise TypeError("M must be an int")
        Li, gain_fix = _stuff_factor(x, L_change)
        xg, ref_dt = _signal(x, self._bc)
        if (xg.size * Li) // M_change == 0:  # ZeroDivisionError for M = 0, like downsample()
            return np.zeros(0, dtype=ref_dt if config.strict_dtype else xg.dtype)
        k = self._kern.get(xg.dtype)
        y = k.updn(xg, Li, M_change, wide=_wide())
        if gain_fix != 1.0:
            y = y * gain_fix
        return _finish(y, ref_dt)

    def freq_resp(self, mode='dB', fs=8000, ylim=[-100, 2]):
        """Plotting helper: out of scope here; delegates to an installed sk_dsp_comm."""
        return _delegate_plot("multirate_FIR", self.b, "freq_resp", mode, fs, ylim)

    def zplane(self, auto_scale=True, size=2, detect_mult=True, tol=0.001):
        return _delegate_plot("multirate_FIR", self.b, "zplane", auto_scale, size, detect_mult, tol)


    # --- extension: block streaming (the reference always restarts from rest) ---
    def filter_stream(self, x, zi=None):
        """(y, zf): filter one block of a longer signal.  zi / zf are the LAST len(b)-1 INPUT samples
        before the block (None = rest), so concatenated block outputs equal filter(whole signal)."""
        xg, ref_dt = _signal(x, self._bc)
        if xg.ndim != 1:
            raise ValueError("filter_stream works on 1-D blocks")
        nh = self.N_forder - 1
        hist = np.zeros(nh, dtype=xg.dtype)
        if zi is not None:
            zi = np.asarray(zi)
            if zi.shape != (nh,):
                raise ValueError("zi must hold the previous %d input samples" % nh)
            hist[:] = zi
        if xg.size == 0:
            return _finish(xg.copy(), ref_dt), hist
        k = self._kern.get(xg.dtype)
        xd = _ffi.DeviceArray(xg.size, xg.dtype, headroom=max(nh, 1))
        yd = _ffi.DeviceArray(xg.size, xg.dtype)
        try:
            xd.write(xg)
            xd.write(hist, at=-nh)
            k.filter_dev(xd, yd, n=xg.size, n_hist=nh)
            y = yd.to_host()
        finally:
            xd.free()
            yd.free()
        zf = np.concatenate([hist, xg])[xg.size:]
        return _finish(y, ref_dt), zf


class multirate_IIR(object):
    """SOS IIR filter / interpolator / decimator (multirate_helper.py:146-208)."""

    def __init__(self, sos):
        self.N_forder = np.sum(np.sign(np.abs(sos[:, 2]))) \
                      + np.sum(np.sign(np.abs(sos[:, 1])))
        self.sos = sos
        log.info('IIR filter order = %d' % self.N_forder)
        self._kern = _KernelCache(self._make, lambda: (self.sos,))

    def _validated(self):
        sos = np.atleast_2d(np.asarray(self.sos))
        if sos.ndim != 2 or sos.shape[1] != 6:
            raise ValueError('sos array must be shape (n_sections, 6)')
        if not (sos[:, 3] == 1).all():
            raise ValueError('sos[:, 3] should be all ones')
        return sos

    def _make(self, code):
        sos = self._validated()
        return [_ffi.IirKernel(code, sos=sos[i:i + _MAX_SOS]) for i in range(0, sos.shape[0], _MAX_SOS)]

    def _prep(self, x):
        sos = self._validated()
        x = np.asarray(x)
        # scipy: dtype = result_type(sos, x); a float32 sos with float32 x stays float32
        ref_dt = np.result_type(sos.dtype, x.dtype, np.float32)
        if ref_dt.kind not in "fc":
            ref_dt = np.dtype(np.float64)
        xg, _ = _signal(x)
        if xg.size == 0:
            raise ValueError("cannot reshape array of size 0 into shape (0)")
        return xg, ref_dt

    def _chain(self, xg, first):
        ks = self._kern.get(xg.dtype)
        y = first(ks[0], xg, _wide() and len(ks) == 1)
        for i, k in enumerate(ks[1:]):
            y = k.filter(y, wide=_wide() and i == len(ks) - 2)
        return y

    def filter(self, x):
        """y = sosfilt(sos, x)  (multirate_helper.py:169-174)"""
        xg, ref_dt = self._prep(x)
        if xg.ndim > 1:
            # N-D: sosfilt filters along the last axis in one call; so does this -- the rows travel as one block and
            # (where the parallel-form scan applies) run as one launch
            ks = self._kern.get(xg.dtype)
            y = xg.reshape(-1, xg.shape[-1])
            step = 1 << 23   # (rows of one launch: below the kernels' 2^24 row limit)
            for i, k in enumerate(ks):
                w = _wide() and i == len(ks) - 1
                if y.shape[0] <= step:
                    y = k.filter_rows(y, wide=w)
                else:
                    y = np.concatenate([k.filter_rows(np.ascontiguousarray(y[r:r + step]), wide=w) for r in range(0, y.shape[0], step)])
            return _finish(y.reshape(xg.shape), ref_dt)
        return _finish(self._chain(xg, lambda k, v, w: k.filter(v, wide=w)), ref_dt)

    # --- extension: block streaming (the reference always restarts from rest) ---
    def filter_stream(self, x, zi=None):
        """(y, zf) == scipy.signal.sosfilt(sos, x, zi=zi): zi / zf have shape (n_sections, 2)
        (complex for a complex signal); None = rest."""
        if np.asarray(x).size:
            xg, ref_dt = self._prep(x)
        else:
            xg, ref_dt = _signal(x)
        if xg.ndim != 1:
            raise ValueError("filter_stream works on 1-D blocks")
        nsec = self._validated().shape[0]
        cplx = np.iscomplexobj(xg)
        z = np.zeros((nsec, 2), dtype=np.complex128 if cplx else np.float64)
        if zi is not None:
            zi = np.asarray(zi)
            if zi.shape != (nsec, 2):
                raise ValueError("zi must have shape (%d, 2)" % nsec)
            if np.iscomplexobj(zi) and not cplx:
                raise ValueError("complex zi needs a complex signal")
            z[...] = zi
        if xg.size == 0:
            return _finish(xg.copy(), ref_dt), z
        zf = np.empty_like(z)
        y = xg
        for i, k in enumerate(self._kern.get(xg.dtype)):
            lo, hi = i * _MAX_SOS, min((i + 1) * _MAX_SOS, nsec)
            part = z[lo:hi]
            flat = np.concatenate([part.real.ravel(), part.imag.ravel()]) if cplx else part.ravel()
            y, out = k.filter_state(y, flat)
            h = (hi - lo) * 2
            zf[lo:hi] = (out[:h] + 1j * out[h:]).reshape(-1, 2) if cplx else out.reshape(-1, 2)
        return _finish(y, ref_dt), zf

    def up(self, x, L_change=12):
        """y = sosfilt(sos, L*upsample(x, L))  (multirate_helper.py:177-183)"""
        Li, gain_fix = _stuff_factor(x, L_change)
        xg, ref_dt = self._prep(x)
        y = self._chain(xg, lambda k, v, w: k.up(v, Li, wide=w))
        if gain_fix != 1.0:
            y = y * gain_fix
        return _finish(y, np.result_type(ref_dt, np.float64))

    def dn(self, x, M_change=12):
        """y = downsample(sosfilt(sos, x), M)  (multirate_helper.py:186-192)"""
        if not isinstance(M_change, int):
            raise TypeError("M must be an int")
        xg, ref_dt = self._prep(x)
        if xg.ndim != 1:
            raise ValueError("cannot reshape array of size %d into shape (%d,%d)"
                             % (xg.size, int(np.floor(len(xg) / M_change)), M_change))
        if len(xg) // M_change == 0:  # fewer than M samples: the reference returns an empty view
            return np.zeros(0, dtype=ref_dt if config.strict_dtype else xg.dtype)
        ks = self._kern.get(xg.dtype)
        y = xg
        for k in ks[:-1]:
            y = k.filter(y)
        return _finish(ks[-1].dn(y, M_change, wide=_wide()), ref_dt)

    def freq_resp(self, mode='dB', fs=8000, ylim=[-100, 2]):
        return _delegate_plot("multirate_IIR", self.sos, "freq_resp", mode, fs, ylim)

    def zplane(self, auto_scale=True, size=2, detect_mult=True, tol=0.001):
        return _delegate_plot("multirate_IIR", self.sos, "zplane", auto_scale, size, detect_mult, tol)


class rate_change(object):
    """Upsample/filter and filter/downsample with an IIR lowpass (multirate_helper.py:45-83).

    The (b, a) design stays on the host with scipy.signal (coefficient generation is
    outside the hot path); lfilter(b, a, .) runs in the GPU scan kernel as a direct-form
    II transposed section of order N, the structure scipy uses."""

    def __init__(self, M_change=12, fcutoff=0.9, N_filt_order=8, ftype='butter'):
        import scipy.signal as signal
        self.M = M_change            # interpolation (.up) / decimation (.dn) factor
        self.fc = fcutoff * .5       # fraction of the post-change Nyquist band kept, in cycles/sample of the low rate
        self.N_forder = N_filt_order
        wn = 2 / self.M * self.fc    # scipy's normalised cutoff: fcutoff / M of the high-rate Nyquist frequency
        if ftype.lower() == 'butter':
            self.b, self.a = signal.butter(self.N_forder, wn)
        elif ftype.lower() == 'cheby1':
            self.b, self.a = signal.cheby1(self.N_forder, 0.05, wn)  # 0.05 dB passband ripple, as the reference fixes it
        else:
            warnings.warn('ftype must be "butter" or "cheby1"')
        self._kern = _KernelCache(lambda code: _ffi.IirKernel(code, b=self.b, a=self.a), lambda: (self.b, self.a))

    def up(self, x):
        """y = lfilter(b, a, M*upsample(x, M))  (multirate_helper.py:69-75)"""
        Li, gain_fix = _stuff_factor(x, self.M)
        b = self.b  # AttributeError for an unsupported ftype, like the reference
        xg, ref_dt = _signal(x)
        if xg.size == 0:
            return np.zeros(0, dtype=ref_dt)
        y = self._kern.get(xg.dtype).up(xg, Li, wide=_wide())
        if gain_fix != 1.0:
            y = y * gain_fix
        return _finish(y, ref_dt)

    def dn(self, x):
        """y = downsample(lfilter(b, a, x), M)  (multirate_helper.py:77-83)"""
        b = self.b
        if not isinstance(self.M, int):
            raise TypeError("M must be an int")
        xg, ref_dt = _signal(x)
        if xg.ndim != 1:
            raise ValueError("cannot reshape array of size %d into shape (%d,%d)"
                             % (xg.size, int(np.floor(len(xg) / self.M)), self.M))
        if len(xg) // self.M == 0:
            return np.zeros(0, dtype=ref_dt)
        return _finish(self._kern.get(xg.dtype).dn(xg, self.M, wide=_wide()), ref_dt)


# ------------------------------------------------------------------ helpers
def _stuff_factor(x, L):
    """upsample()'s argument checks + its integer stuffing factor int(L-1)+1.

    The reference multiplies the zero-stuffed signal by L itself (possibly non-integer)
    while stuffing int(L-1)+1: return the integer factor and the residual gain."""
    if not hasattr(x, "reshape"):
        raise AttributeError("'%s' object has no attribute 'reshape'" % type(x).__name__)
    if x.ndim != 1:
        raise ValueError("cannot reshape array of size %d into shape (%d,1)" % (x.size, len(x)))
    Lz = int(L - 1)
    if Lz < 0:
        raise ValueError("negative dimensions are not allowed")
    Li = Lz + 1
    return Li, float(L) / float(Li)


def _delegate_plot(cls_name, coeffs, method, *args):
    try:
        import sk_dsp_comm.multirate_helper as ref
    except Exception:
        raise NotImplementedError("%s.%s is a matplotlib helper outside the accelerated path; "
                                  "install scikit-dsp-comm to use it" % (cls_name, method))
    return getattr(getattr(ref, cls_name)(coeffs), method)(*args)
