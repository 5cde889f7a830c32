"""Rate-change and CIC-tap primitives with the signatures of sk_dsp_comm.sigsys.

  upsample(x, L)       /root/reference/src/sk_dsp_comm/sigsys.py:3031-3053
  downsample(x, M, p)  /root/reference/src/sk_dsp_comm/sigsys.py:3056-3083
  cic(m, k)            /root/reference/src/sk_dsp_comm/sigsys.py:62-93

  os_filter(x,h,N,mode) /root/reference/src/sk_dsp_comm/sigsys.py:482-540   ("next" row)
  oa_filter(x,h,N,mode) /root/reference/src/sk_dsp_comm/sigsys.py:543-598

Callers around the hot path, same signatures, the filtering on the GPU:
  interp24(x) / deci24(x)         sigsys.py:2945-3028   3-stage (b,a) Butterworth x24 / /24
  ten_band_eq_filt(x, GdB, Q)     sigsys.py:96-141      ten peaking biquads -> one SOS scan
  peaking(GdB, fc, Q, fs)         sigsys.py:202-260     (host design)
  rc_imp / sqrt_rc_imp            sigsys.py:1847-1945   (host pulse design)
  nrz_bits / nrz_bits2            sigsys.py:2120-2211   lfilter(b, 1, zero-stuffed data) -> polyphase .up
  fft_filt_bank(x, h, ...)        sigsys.py:2588-2694   (8f-4) one FIR per band with frequency-shifted taps

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
    box = np.ones(m)
    taps = box
    for _ in range(1, k):                 # k <= 1 (0 included) leaves a single boxcar, as in the reference
        taps = np.convolve(taps, box)     # exact: the entries are integer path counts far below 2^53
    return taps / np.sum(taps)


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
    if config.strict_dtype and xg.dtype != out_dt:
        xg = xg.astype(out_dt)  # widen the n inputs, not the n*L outputs: the move itself is exact
    return _ffi.upsample(xg, Li)


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
    x = np.asarray(x)
    if len(x) == 0:
        return (np.zeros(0), np.zeros((0, 0))) if mode == 1 else np.zeros(0)
    y = mrh.multirate_FIR(np.asarray(h))._filter(x, True)   # the reference's float64 / complex128 whatever config.strict_dtype says
    y = np.ascontiguousarray(np.real(y), dtype=np.float64)
    if mode == 1:
        return y, _frame_matrix(x, np.asarray(h), int(N), name)
    return y


def _frame_matrix(x, h, N, name):
    """The diagnostic matrix of os_filter / oa_filter(mode=1): row k holds the N outputs of frame k's circular
    convolution at the frame's position (sigsys.py:517-540, 582-598).  It is a teaching aid of the reference's frame
    loop (Nframe x Nx float64 -- quadratic in the signal length), computed here on the host with one batched FFT;
    the filtered signal itself comes from the GPU."""
    P, Nx0 = len(h), len(x)
    L = N - P + 1
    H = np.fft.fft(h, N)
    if name == "os_filter":
        xp = np.concatenate([np.zeros(P - 1), x])
        Nx = len(xp)
        nframe = int(np.ceil(Nx / float(L)))
        xp = np.concatenate([xp, np.zeros(nframe * L - Nx + N)])
        frames = np.stack([xp[k * L:k * L + N] for k in range(nframe)])
    else:
        Nx = Nx0
        nframe = int(np.ceil(Nx / float(L)))
        xp = np.concatenate([x, np.zeros(nframe * L - Nx)])
        frames = xp.reshape(nframe, L)
    yk = np.real(np.fft.ifft(np.fft.fft(frames, N, axis=1) * H, axis=1))
    y_mat = np.zeros((nframe, nframe * N))
    for k in range(nframe):
        y_mat[k, k * L:k * L + N] = yk[k]
    return y_mat[:, P - 1:Nx] if name == "os_filter" else y_mat[:, 0:Nx]


def os_filter(x, h, N, mode=0):
    """Overlap-and-save FIR filtering (sigsys.py:482-540): y = real(lfilter(h, 1, x))."""
    return _transform_domain_fir(x, h, N, mode, "os_filter")


def oa_filter(x, h, N, mode=0):
    """Overlap-and-add FIR filtering (sigsys.py:543-598): y = real(lfilter(h, 1, x))."""
    return _transform_domain_fir(x, h, N, mode, "oa_filter")


# ---------------------------------------------------------------------------------------------
# callers around the hot path
# ---------------------------------------------------------------------------------------------
def _lfilter_dtype(x):
    """(device array, result dtype) for scipy.signal.lfilter(b, a, x) with float64 coefficients."""
    from . import multirate_helper as mrh
    return mrh._signal(np.asarray(x))


def _butter_stage(order, wn, dtype):
    import scipy.signal as signal
    b, a = signal.butter(order, wn)
    return _ffi.IirKernel(_ffi.code_of(dtype), b=b, a=a)


def interp24(x):
    """Interpolate by 24 in three stages x2, x3, x4, each lfilter(b, a, L*upsample(., L)) with a
    10th-order Butterworth at 1/L (sigsys.py:2945-2985).  One fused zero-stuff + scan per stage."""
    from . import multirate_helper as mrh
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("cannot reshape array of size %d into shape (%d,1)" % (x.size, len(x)))
    xg, ref_dt = _lfilter_dtype(x)
    if xg.size == 0:
        return np.zeros(0, dtype=ref_dt)
    y = xg
    for L in (2, 3, 4):
        y = _butter_stage(10, 1.0 / L, y.dtype).up(y, L, wide=config.strict_dtype and L == 4)
    return mrh._finish(y, ref_dt)


def deci24(x):
    """Decimate by 24 in three stages /2, /3, /4, each downsample(lfilter(b, a, .), M) with a
    10th-order Butterworth at 1/M (sigsys.py:2988-3028); only the kept samples are written."""
    from . import multirate_helper as mrh
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("deci24 expects a 1-D signal")
    xg, ref_dt = _lfilter_dtype(x)
    y = xg
    for M in (2, 3, 4):
        if len(y) // M == 0:
            return np.zeros(0, dtype=ref_dt if config.strict_dtype else xg.dtype)
        y = _butter_stage(10, 1.0 / M, y.dtype).dn(y, M, wide=config.strict_dtype and M == 4)
    return mrh._finish(y, ref_dt)


def peaking(GdB, fc, Q=3.5, fs=44100.):
    """Second-order peaking (bell) equaliser section, gain GdB at fc (sigsys.py:202-260).
    Returns (b, a) with a[0] = 1."""
    mu = 10.0 ** (GdB / 20.0)
    w0 = 2.0 * np.pi * fc / fs
    kq = 4.0 / (1.0 + mu) * np.tan(w0 / (2.0 * Q))
    c0 = np.cos(w0)
    den_b, den_a = 1.0 + kq * mu, 1.0 + kq
    b = (den_b / den_a) * np.array([1.0, -2.0 * c0 / den_b, (1.0 - kq * mu) / den_b])
    a = np.array([1.0, -2.0 * c0 / den_a, (1.0 - kq) / den_a])
    return b, a


def ten_band_eq_filt(x, GdB, Q=3.5):
    """Ten octave-spaced peaking filters (31.25 Hz ... 16 kHz at fs = 44.1 kHz) in cascade
    (sigsys.py:96-141).  The reference runs ten lfilter passes; here the ten biquads are one
    SOS cascade in a single exact scan."""
    from . import multirate_helper as mrh
    nb = len(GdB)
    if not nb == 10:
        raise ValueError("GdB length not equal to ten")
    fc = 31.25 * 2.0 ** np.arange(nb)
    sos = np.zeros((nb, 6))
    for k in range(nb):
        sos[k, :3], sos[k, 3:] = peaking(GdB[k], fc[k], Q)
    xg, ref_dt = _lfilter_dtype(x)
    if xg.size == 0:
        return np.zeros(0)
    y = _ffi.IirKernel(_ffi.code_of(xg.dtype), sos=sos).filter(xg, wide=config.strict_dtype)
    return mrh._finish(y, ref_dt)


def rc_imp(Ns, alpha, M=6):
    """Truncated raised-cosine pulse, 2*M*Ns+1 samples (sigsys.py:1847-1891)."""
    n = np.arange(-M * Ns, M * Ns + 1)
    t = n / float(Ns)
    den = 1.0 - 4.0 * (alpha * t) ** 2
    sing = den == 0
    b = np.sinc(t) * np.cos(np.pi * alpha * t) / np.where(sing, 1.0, den)
    if np.any(sing):
        b[sing] = np.pi / 4.0 * np.sinc(1.0 / (2.0 * alpha))
    return b


def sqrt_rc_imp(Ns, alpha, M=6):
    """Truncated square-root raised-cosine pulse, 2*M*Ns+1 samples (sigsys.py:1894-1945)."""
    n = np.arange(-M * Ns, M * Ns + 1)
    t = n / float(Ns)
    a = alpha
    den = 1.0 - 16.0 * a ** 2 * t ** 2
    sing = np.abs(den) <= np.finfo(np.float32).eps / 2
    b = 4.0 * a / (np.pi * np.where(sing, 1.0, den)) * (np.cos((1.0 + a) * np.pi * t)
                                                    + np.sinc((1.0 - a) * t) * (1.0 - a) * np.pi / (4.0 * a))
    if np.any(sing):
        b[sing] = 0.5 * ((1.0 + a) * np.sin((1.0 + a) * np.pi / (4.0 * a))
                         - (1.0 - a) * np.cos((1.0 - a) * np.pi / (4.0 * a))
                         + (4.0 * a) / np.pi * np.sin((1.0 - a) * np.pi / (4.0 * a)))
    return b


def _pulse(pulse, ns, alpha, m, err='pulse type must be rec, rc, or src'):
    kind = pulse.lower()
    if kind == 'rect':
        return np.ones(int(ns))
    if kind == 'rc':
        return rc_imp(ns, alpha, m)
    if kind == 'src':
        return sqrt_rc_imp(ns, alpha, m)
    raise ValueError(err)


def pulse_shape(symbols, b, ns):
    """lfilter(b, 1, upsample(symbols, ns)) -- the pulse-shaping step of nrz_bits*, the *_bb
    transmitters of digitalcom (digitalcom.py:1676, 1821) -- as ONE polyphase interpolation:
    only the ns-th of the products that do not multiply a stuffed zero is computed."""
    from . import multirate_helper as mrh
    ns = int(ns)
    sym = np.asarray(symbols)
    if sym.size == 0:
        return np.zeros(0, dtype=np.complex128 if np.iscomplexobj(sym) else np.float64)
    if ns == 1:
        return mrh.multirate_FIR(np.asarray(b, dtype=np.float64)).filter(sym)
    # multirate_FIR.up applies the interpolation gain ns to the stuffed signal; lfilter here does not
    return mrh.multirate_FIR(np.asarray(b, dtype=np.float64) / ns).up(sym.astype(np.result_type(sym.dtype, np.float64)), ns)


def nrz_bits2(data, Ns, pulse='rect', alpha=0.25, M=6):
    """NRZ +-1 waveform from user bits with pulse shaping (sigsys.py:2163-2211): (x, b/Ns)."""
    data = np.asarray(data)
    b = _pulse(pulse, Ns, alpha, M)
    x = pulse_shape(2 * data - 1, b, Ns)
    return x, b / float(Ns)


def nrz_bits(n_bits, ns, pulse='rect', alpha=0.25, m=6):
    """NRZ +-1 waveform from random bits (sigsys.py:2120-2160): (x, b/ns, data)."""
    data = np.random.randint(0, 2, n_bits)
    x, b = nrz_bits2(data, ns, pulse, alpha, m)
    return x, b, data


def env_det(x):
    """Ideal envelope detector: half-wave rectifier (sigsys.py:2913-2942)."""
    x = np.asarray(x)
    return np.where(x >= 0, x, 0).astype(np.float64)


def am_tx(m, a_mod, fc=75e3):
    """AM transmitter of the Chapter-17 case study (sigsys.py:2784-2819): the message is interpolated by 24 on the GPU
    (interp24), the carrier is applied on the host.  Returns (x192, t192, m24)."""
    m24 = interp24(m)
    t192 = np.arange(len(m24)) / 192.0e3
    m_max = np.max(np.abs(m24))
    x192 = (1 + a_mod * m24 / m_max) * np.cos(2 * np.pi * fc * t192)
    return x192, t192, m24


def am_rx(x192):
    """AM envelope-detector receiver (sigsys.py:2822-2866): env_det -> deci24 (GPU) -> DC removal, plus the 192 ksps
    monitor output through two passes of a 5th-order Butterworth at 5 kHz (one 10th-order cascade launch here).
    Returns (m_rx8, t8, m_rx192, x_edet192)."""
    import scipy.signal as signal
    x_edet192 = env_det(x192)
    m_rx8 = deci24(x_edet192)
    m_rx8 = m_rx8 - np.mean(m_rx8)
    t8 = np.arange(len(m_rx8)) / 8.0e3
    b192, a192 = signal.butter(5, 2 * 5.0e3 / 192.0e3)
    if len(x_edet192):
        sos = _ffi.tf2sos(b192, a192)
        from . import multirate_helper as mrh
        m_rx192 = mrh.multirate_IIR(np.vstack([sos, sos])).filter(x_edet192)   # lfilter(b, a, lfilter(b, a, .))
    else:
        m_rx192 = np.zeros(0)
    m_rx192 = m_rx192 - np.mean(m_rx192) if len(m_rx192) else m_rx192
    return m_rx8, t8, m_rx192, x_edet192


def fft_filt_bank(x_in, h_filt, n_fft2=512, n_bands2=0, bs=0.2, fs=1.0, n_band_odd=True):
    """Streaming filter bank of 2*n_bands2+1 (or 2*n_bands2) bands spaced by ~bs Hz
    (sigsys.py:2588-2694).  The reference overlap-saves with a 2*n_fft2 FFT and rolls H by whole
    bins per band; rolling H by s bins IS the filter h[n]*exp(j*2*pi*s*n/(2*n_fft2)), so every band
    is one complex-tap FIR over the same device-resident input (fir_ols.hip / fir_direct.hip).
    Like the reference only whole blocks of n_fft2 samples are produced; the tail stays zero.
    Returns (y_filt_bank[bands, len(x)], freq_axis, freq_axis_desired)."""
    h_filt = np.asarray(h_filt)
    if len(h_filt) > n_fft2:
        raise ValueError('Error: Must have Nfft2 = %d >= %d = len(h_ref)' % (n_fft2, len(h_filt)))
    x_in = np.asarray(x_in)
    n_x = len(x_in)
    if n_band_odd:
        n_tot = 2 * n_bands2 + 1
        step = int(round(bs * 2 * n_fft2 / fs))
        shifts = [j * step - n_bands2 * step for j in range(n_tot)]
    else:
        n_tot = 2 * n_bands2
        step = int(round(bs / 2 * 2 * n_fft2 / fs))
        shifts = [j * 2 * step - (2 * n_bands2 - 1) * step for j in range(n_tot)]
    bs_actual = step * fs / (2 * n_fft2)
    print('N_band_step = %d' % (step,))
    print('N_band_step = %d and BS_Hz_actual = %3.2f Hz' % (step, bs_actual))
    print('N_bands_tot = %d and span_Hz = +/- %4.2f Hz' % (n_tot, n_bands2 * bs_actual))
    y = np.zeros((n_tot, n_x), dtype=complex)
    n_use = (n_x // n_fft2) * n_fft2
    if n_use and n_tot:
        single = x_in.dtype in (np.float32, np.complex64) and not config.strict_dtype
        cdt = np.complex64 if single else np.complex128
        # one device-resident input; the bands in batches of rows of ONE device block (2 GiB at most), one copy back per batch
        xd = _ffi.DeviceArray.from_host(np.ascontiguousarray(x_in[:n_use], dtype=cdt))
        per = max(1, min(n_tot, (2 << 30) // max(n_use * np.dtype(cdt).itemsize, 1)))
        yd = _ffi.DeviceArray(n_use * per, cdt)
        try:
            n = np.arange(len(h_filt))
            for j0 in range(0, n_tot, per):
                rows = min(per, n_tot - j0)
                for j in range(rows):
                    taps = h_filt * np.exp(2j * np.pi * shifts[j0 + j] * n / (2 * n_fft2))
                    _ffi.FirKernel(taps, _ffi.code_of(cdt)).filter_dev(xd, yd.window(j * n_use, n_use))
                y[j0:j0 + rows, :n_use] = yd.to_host(0, rows * n_use).reshape(rows, n_use)
        finally:
            xd.free()
            yd.free()
    if n_band_odd:
        freq_axis = np.arange(-n_bands2 * step, n_bands2 * step + step, step) * fs / 2 / n_fft2
        freq_axis_desired = np.rint(freq_axis / bs) * bs
    else:
        freq_axis = np.arange(-(2 * n_bands2 - 1) * step, n_bands2 * step + 2 * (step + 1), 2 * step) * fs / 2 / n_fft2
        freq_axis_desired = np.rint(freq_axis / (bs / 2)) * (bs / 2)
    return y, freq_axis, freq_axis_desired
