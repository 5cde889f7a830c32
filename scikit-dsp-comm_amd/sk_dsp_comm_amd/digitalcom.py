"""Gray-coded QAM / MPSK baseband transmitters with the signatures of sk_dsp_comm.digitalcom
(SURVEY.md 8f-2).  Symbol mapping is vectorised NumPy on the host; the expensive step,
lfilter(b, 1, upsample(x_IQ, ns)) (digitalcom.py:1676, 1821), is one polyphase interpolation on
the GPU (sigsys.pulse_shape -> multirate_FIR.up -> fir_direct.hip / fir_ols.hip).

  qam_gray_encode_bb(n_symb, ns, mod, pulse, alpha, m_span, ext_data)   digitalcom.py:1584-1681
  mpsk_gray_encode_bb(n_symb, ns, mod, pulse, alpha, m_span, ext_data)  digitalcom.py:1742-1826
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
