"""Run-time switches of the drop-in layer.

strict_dtype (default True): return exactly the dtypes the reference returns
    (float64 / complex128 from lfilter/sosfilt/upsample, SURVEY.md 7.3 "drop-in dtype
    semantics") by up-casting the float32 / complex64 GPU result once on the host.
    Set False to keep the native GPU precision class and skip that pass.
"""
import os

strict_dtype = os.environ.get("SKDSP_STRICT_DTYPE", "1") not in ("0", "false", "False")
