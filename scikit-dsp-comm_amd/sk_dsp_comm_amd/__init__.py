"""sk_dsp_comm_amd -- MI355X-native streaming FIR/IIR/multirate filters behind the
scikit-dsp-comm API (multirate_helper.rate_change / multirate_FIR / multirate_IIR and
sigsys.upsample / downsample / cic).  Host code is Python + ctypes; the arithmetic runs in
hand-written HIP kernels for gfx950 (libskdsp_hip.so).  No PyTorch, no CPU fallback.
"""
from . import _ffi
from . import config
from . import sigsys
from . import multirate_helper
from . import digitalcom
from .multirate_helper import rate_change, multirate_FIR, multirate_IIR
from .sigsys import upsample, downsample, cic

__version__ = "0.1.0"
__all__ = ["sigsys", "multirate_helper", "digitalcom", "rate_change", "multirate_FIR", "multirate_IIR", "upsample", "downsample",
           "cic", "config", "install"]


def install():
    """Optional: patch an installed `sk_dsp_comm` so existing code picks up the GPU path."""
    import sk_dsp_comm.multirate_helper as ref_mrh
    import sk_dsp_comm.sigsys as ref_ss
    for name in ("rate_change", "multirate_FIR", "multirate_IIR"):
        setattr(ref_mrh, name, getattr(multirate_helper, name))
    for name in ("upsample", "downsample", "cic"):
        setattr(ref_ss, name, getattr(sigsys, name))
