"""Coefficient header I/O with the file formats of sk_dsp_comm.coeff2header, so
that taps written by the reference (or for CMSIS-DSP targets) drop straight into the GPU filter
objects and vice versa.  Host-side text; nothing here touches the device.

  fir_header(fname_out, h)          float32_t h_FIR[M_FIR], 3 per line, %15.12f   coeff2header.py:42-74
  fir_fix_header(fname_out, h)      int16_t  h_FIR[M_FIR], Q15, 8 per line, %5d   coeff2header.py:77-110
  iir_sos_header(fname_out, sos)    float32_t ba_coeff[5*STAGES]: b0,b1,b2,-a1,-a2 coeff2header.py:113-155
  read_fir_header(fname)            -> taps (float64; Q15 files are scaled back by 2**-15)
  read_sos_header(fname)            -> sos (n_sections, 6) with a0 = 1
"""
import re

import numpy as np

_RULE_FIR = '/************************************************************************/\n'
_RULE_SOS = '/*********************************************************/\n'


def _array_body(tokens, per_line, indent):
    """`per_line` comma-separated tokens per row, continuation rows indented."""
    rows = [','.join(tokens[i:i + per_line]) for i in range(0, len(tokens), per_line)]
    return (',\n' + ' ' * indent).join(rows)


def _fir_text(decl, tokens, per_line, indent):
    return ('//define a FIR coefficient Array\n\n'
            '#include <stdint.h>\n\n'
            '#ifndef M_FIR\n'
            '#define M_FIR %d\n'
            '#endif\n' % len(tokens)
            + _RULE_FIR +
            '/*                         FIR Filter Coefficients                      */\n'
            + decl + _array_body(tokens, per_line, indent) + '};\n' + _RULE_FIR)


def fir_header(fname_out, h):
    """Write a float FIR coefficient header (coeff2header.py:42-74)."""
    with open(fname_out, 'wt') as f:
        f.write(_fir_text('float32_t h_FIR[M_FIR] = {', ['%15.12f' % v for v in h], 3, 26))


def fir_fix_header(fname_out, h):
    """Write a Q15 fixed-point FIR coefficient header (coeff2header.py:77-110)."""
    hq = np.int16(np.rint(np.asarray(h) * 2 ** 15))
    with open(fname_out, 'wt') as f:
        f.write(_fir_text('int16_t h_FIR[M_FIR] = {', ['%5d' % v for v in hq], 8, 24))


def iir_sos_header(fname_out, SOS_mat):
    """Write a CMSIS-DSP style SOS header: per stage b0, b1, b2, -a1, -a2 (coeff2header.py:113-155)."""
    sos = np.asarray(SOS_mat)
    ns = sos.shape[0]
    stages = ['    %+-13e, %+-13e, %+-13e,\n    %+-13e, %+-13e' % (r[0], r[1], r[2], -r[4], -r[5]) for r in sos]
    with open(fname_out, 'wt') as f:
        f.write('//define a IIR SOS CMSIS-DSP coefficient array\n\n'
                '#include <stdint.h>\n\n'
                '#ifndef STAGES\n'
                '#define STAGES %d\n'
                '#endif\n' % ns
                + _RULE_SOS +
                '/*                     IIR SOS Filter Coefficients       */\n'
                'float32_t ba_coeff[%d] = { //b0,b1,b2,a1,a2,... by stage\n' % (5 * ns)
                + ',\n'.join(stages) + '\n};\n' + _RULE_SOS)


def _initializer(text, array_name):
    m = re.search(r'(\w+)\s+' + array_name + r'\s*\[[^\]]*\]\s*=\s*\{(.*?)\}\s*;', text, re.S)
    if not m:
        raise ValueError("no %s[...] = {...}; initializer found" % array_name)
    body = re.sub(r'//[^\n]*', '', m.group(2))
    vals = [v for v in re.split(r'[,\s]+', body) if v]
    return m.group(1), np.array([float(v) for v in vals])


def read_fir_header(fname):
    """Taps from a fir_header / fir_fix_header file (Q15 integers are scaled by 2**-15)."""
    ctype, vals = _initializer(open(fname).read(), 'h_FIR')
    return vals * 2.0 ** -15 if ctype == 'int16_t' else vals


def read_sos_header(fname):
    """SOS matrix (n_sections, 6) from an iir_sos_header file (stored as b0,b1,b2,-a1,-a2)."""
    _, vals = _initializer(open(fname).read(), 'ba_coeff')
    if vals.size % 5:
        raise ValueError("ba_coeff must hold 5 values per stage")
    ba = vals.reshape(-1, 5)
    sos = np.ones((ba.shape[0], 6))
    sos[:, :3] = ba[:, :3]
    sos[:, 4:] = -ba[:, 3:]
    return sos
