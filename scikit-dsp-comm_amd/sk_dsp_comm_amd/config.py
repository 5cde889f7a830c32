"""Run-time switches of the drop-in layer.

strict_dtype (default True): return exactly the dtypes the reference returns (float64 / complex128 from
    lfilter / sosfilt / upsample, SURVEY.md 7.3 "drop-in dtype semantics"); float32 / complex64 kernel results are
    widened on the device before they are copied back.  False keeps the kernel's own precision class.

precision (default "input"): which arithmetic the filters run in.
    "input"   float32 / complex64 / float16 arrays are filtered in float32 (float64 state inside the IIR scan): results
              within 1e-6 of the reference (the accuracy contract of BASELINE.json; the reference itself promotes such
              inputs to float64, so a result typed float64 by strict_dtype still carries float32 accuracy);
              float64 / complex128 / integer arrays are filtered in float64.
    "double"  everything in float64, like the reference: 1e-12 agreement, also for attenuated (stop-band) outputs
              where float32 arithmetic only holds a bound relative to the INPUT.  Long FIRs run
              on the float64 overlap-save tile (csrc/fir_ols64.hip): about twice the time of float32, as their bytes are.
    "single"  everything in float32, also float64 inputs: the fast kernels for NumPy's default dtype when 1e-6 is enough.
"""
import os

strict_dtype = os.environ.get("SKDSP_STRICT_DTYPE", "1") not in ("0", "false", "False")
precision = os.environ.get("SKDSP_PRECISION", "input")
