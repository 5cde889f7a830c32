"""Gray-coded QAM / MPSK baseband transmitters with the signatures of sk_dsp_comm.digitalcom
(SURVEY.md 8f-2).  Symbol mapping is vectorised NumPy on the host; the expensive step,
lfilter(b, 1, upsample(x_IQ, ns)) (digitalcom.py:1676, 1821), is one polyphase interpolation on
the GPU (sigsys.pulse_shape -> multirate_FIR.up -> fir_direct.hip / fir_ols.hip).

  qam_gray_encode_bb(n_symb, ns, mod, pulse, alpha, m_span, ext_data)   digitalcom.py:1584-1681
  mpsk_gray_encode_bb(n_symb, ns, mod, pulse, alpha, m_span, ext_data)  digitalcom.py:1742-1826
  qam_bb, mpsk_bb, gmsk_bb, rz_bits, time_delay (constant delay)         digitalcom.py:418-492, 613-667, 585-610, 998-1048, 1089-1131
"""
import numpy as np

from .sigsys import upsample, downsample, cic, rc_imp, sqrt_rc_imp, pulse_shape, _pulse  # noqa: F401  (re-exported like digitalcom.py:40-47)


def _word_values(data, width):
    """MSB-first integer value of consecutive `width`-bit words."""
    w = 2 ** np.arange(width - 1, -1, -1)
    return np.asarray(data).reshape(-1, width) @ w


def _gray_lut(width):
    """The reference's bin2gray tables: entry k is the running XOR of k's bits from the MSB down."""
    k = np.arange(2 ** width)
    v = k.copy()
    shift = 1
    while shift < width:
        v ^= v >> shift
        shift *= 2
    return v


def _take_bits(n_symb, bits_per_symbol, ext_data):
    if n_symb is None:
        n_symb = int(np.floor(len(ext_data) / bits_per_symbol))
        data = np.asarray(ext_data)[:n_symb * bits_per_symbol]
    else:
        data = np.random.randint(0, 2, size=bits_per_symbol * n_symb)
    return n_symb, data


def _shape(x_iq, ns, pulse, alpha, m_span):
    b = _pulse(pulse, ns, alpha, m_span, err='pulse shape must be src, rc, or rect')
    return pulse_shape(x_iq, b, ns), b / sum(b)


def qam_gray_encode_bb(n_symb, ns, mod=4, pulse='rect', alpha=0.35, m_span=6, ext_data=None):
    """Gray-mapped square QAM complex baseband transmitter: (x, b, tx_data)."""
    if mod not in (2, 4, 16, 64, 256):
        raise ValueError('M must be 2, 4, 16, 64, 256')
    bps = int(np.log2(mod))
    n_symb, data = _take_bits(n_symb, bps, ext_data)
    x_m = np.sqrt(mod) - 1
    if mod == 2:  # BPSK special case
        x_iq = 2 * data - 1
        x_m = 1
    else:
        half = bps // 2
        words = _word_values(data, half).reshape(n_symb, 2)  # [I word, Q word] per symbol, MSB first
        lut = _gray_lut(half)
        x_iq = (2 * lut[words[:, 0]] - x_m) + 1j * (2 * lut[words[:, 1]] - x_m)
    if ns > 1:
        x, b = _shape(x_iq, ns, pulse, alpha, m_span)
        return x / x_m, b, data
    return x_iq / x_m, 1, data


def mpsk_gray_encode_bb(n_symb, ns, mod=4, pulse='rect', alpha=0.35, m_span=6, ext_data=None):
    """Gray-mapped M-PSK complex baseband transmitter: (x, b, tx_data)."""
    if mod not in (2, 4, 8, 16, 32):
        raise ValueError('M must be 2, 4, 8, 16, or 32')
    bps = int(np.log2(mod))
    n_symb, data = _take_bits(n_symb, bps, ext_data)
    if mod == 2:
        x_iq = 2 * data - 1
    else:
        idx = _gray_lut(bps)[_word_values(data, bps)]
        phase = 2 * np.pi * idx / mod + (np.pi / mod if mod == 4 else 0.0)
        x_iq = np.exp(1j * phase)
    if ns > 1:
        x, b = _shape(x_iq, ns, pulse, alpha, m_span)
        return x, b, data
    return x_iq, 1, data


# ---- the remaining lfilter(b, 1, .) callers of digitalcom.py (488, 608, 666, 1047, 1130) ------------------------------
def qam_bb(n_symb, ns, mod='16qam', pulse='rect', alpha=0.35):
    """Square-QAM complex baseband transmitter without Gray mapping (digitalcom.py:418-492): (x, b, tx_data).  The
    random symbols are drawn exactly as the reference draws them (I levels, then Q levels)."""
    b = _pulse(pulse, ns, alpha, 6, err='pulse shape must be src, rc, or rect')
    levels = {'qpsk': 2, '16qam': 4, '64qam': 8, '256qam': 16}.get(mod.lower())
    if levels is None:
        raise ValueError('Unknown mod_type')
    x_i = 2 * np.random.randint(0, levels, n_symb) - (levels - 1)
    x_q = 2 * np.random.randint(0, levels, n_symb) - (levels - 1)
    symb = (x_i + 1j * x_q).astype(np.complex128)
    if levels > 2:
        symb = symb / (levels - 1)
    x = pulse_shape(symb, b, ns) if n_symb else np.zeros(0, dtype=np.complex128)
    return x, b / sum(b), x_i + 1j * x_q


def mpsk_bb(n_symb, ns, mod, pulse='rect', alpha=0.25, m=6):
    """M-ary PSK complex baseband transmitter (digitalcom.py:613-667): (x, b / ns, data)."""
    data = np.random.randint(0, mod, n_symb)
    xs = np.exp(1j * 2 * np.pi / mod * data)
    b = _pulse(pulse, ns, alpha, m, err='pulse type must be rec, rc, or src')
    x = pulse_shape(xs, b, ns) if n_symb else np.zeros(0, dtype=np.complex128)
    if mod == 4:
        x = x * np.exp(1j * np.pi / 4)
    return x, b / float(ns), data


def rz_bits(n_bits, ns, pulse='rect', alpha=0.25, m=6):
    """Return-to-zero 0/1 waveform from random bits (digitalcom.py:998-1048): (x, b / ns, data)."""
    data = np.random.randint(0, 2, n_bits)
    try:
        b = _pulse(pulse, ns, alpha, m, err='pulse type must be rec, rc, or src')
    except ValueError as e:
        # the reference only warns here (digitalcom.py:1045-1046) and then fails on its unassigned `b` (:1047)
        import warnings
        warnings.warn(str(e))
        raise UnboundLocalError("local variable 'b' referenced before assignment")
    x = pulse_shape(data, b, ns) if n_bits else np.zeros(0)
    return x, b / float(ns), data


def gmsk_bb(n_bits, ns, msk=0, bt=0.35):
    """MSK / GMSK complex baseband transmitter (digitalcom.py:585-610): (y, data).  The NRZ shaping and the Gaussian
    pre-modulation filter (8 ns + 1 taps) run on the GPU; the phase accumulation is a host cumsum as in the reference."""
    from .sigsys import nrz_bits
    from . import multirate_helper as mrh
    x, b, data = nrz_bits(n_bits, ns)
    span = 4
    n = np.arange(-span * ns, span * ns + 1)
    p = np.exp(-2 * np.pi ** 2 * bt ** 2 / np.log(2) * (n / float(ns)) ** 2)
    p = p / np.sum(p)
    if msk != 0:
        x = mrh.multirate_FIR(p).filter(x)
    y = np.exp(1j * np.pi / 2 * np.cumsum(x) / ns)
    return y, data


def time_delay(x, d, n=4):
    """Farrow-structure time delay (digitalcom.py:1089-1160).  A Python float / int d is the reference's constant-delay
    branch (:1110-1131): cubic Lagrange taps behind fix(d) whole samples, ONE lfilter call -- filtered on the GPU.
    Anything else is the time-varying branch (:1132-1160, d[k] per sample): a 4-tap gather with polynomial weights,
    y[k] = ((v3 mu + v2) mu + v1) mu + v0,  v_j = W_j . x[k-Nd_k+1 .. k-Nd_k-2],  mu = 1 - frac(d[k]),
    evaluated for all k at once on the host (the reference loops over k in Python; there is no filter call to
    accelerate: every output has its own taps)."""
    from . import multirate_helper as mrh
    if type(d) == float or type(d) == int:   # (the reference's own test, :1110)
        if int(np.fix(d)) < 1 or int(np.fix(d)) > n - 2:
            raise ValueError("time_delay: the integer part of d must lie in [1, n - 2]")
        frac = d - np.fix(d)
        nd = int(np.fix(d))
        b = np.zeros(nd + 4)
        b[nd] = -(frac - 1) * (frac - 2) * (frac - 3) / 6.
        b[nd + 1] = frac * (frac - 2) * (frac - 3) / 2.
        b[nd + 2] = -frac * (frac - 1) * (frac - 3) / 2.
        b[nd + 3] = frac * (frac - 1) * (frac - 2) / 6.
        return mrh.multirate_FIR(b).filter(np.asarray(x))
    x = np.asarray(x)
    d = np.asarray(d, dtype=np.float64)
    if d.ndim == 0:                           # a NumPy scalar: the reference would index it and fail; treat it as constant per sample
        d = np.full(len(x), float(d))
    if len(d) < len(x):
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (len(d), len(d)))   # what d[k] raises in the reference
    d = d[:len(x)]
    if len(x) and (np.fix(np.min(d)) < 1 or np.fix(np.max(d)) > n - 2):
        raise ValueError("time_delay: the integer part of d must lie in [1, n - 2]")
    nd = np.fix(d).astype(np.int64)
    mu = 1.0 - (d - np.fix(d))
    k = np.arange(len(x))
    xr = np.real(x).astype(np.float64) if np.iscomplexobj(x) else x.astype(np.float64)   # y = np.zeros(len(x)): real (:1139)
    taps = np.zeros((4, len(x)))
    for i in range(4):                        # X[Nd-1+i] = x[k - (Nd-1) - i], zero before the start
        idx = k - (nd - 1) - i
        ok = idx >= 0
        taps[i, ok] = xr[idx[ok]]
    w3 = np.array([1. / 6, -1. / 2, 1. / 2, -1. / 6])
    w2 = np.array([0, 1. / 2, -1., 1. / 2])
    w1 = np.array([-1. / 6, 1., -1. / 2, -1. / 3])
    w0 = np.array([0, 0, 1., 0])
    v3, v2, v1, v0 = w3 @ taps, w2 @ taps, w1 @ taps, w0 @ taps
    return ((v3 * mu + v2) * mu + v1) * mu + v0
