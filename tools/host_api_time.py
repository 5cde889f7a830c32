import sys, time, numpy as np
sys.path.insert(0,'scikit-dsp-comm_amd'); sys.path.insert(0,'.')
import bench
from sk_dsp_comm_amd import multirate_helper as mrh, config
b=bench.firwin_lowpass(1024,0.2)
f=mrh.multirate_FIR(b)
rng=np.random.default_rng(0)
n=1<<24
x=((rng.standard_normal(n)+1j*rng.standard_normal(n))/np.sqrt(2)).astype(np.complex64)
f.filter(x[:100000])
for strict in (True, False):
    config.strict_dtype=strict
    t0=time.perf_counter(); y=f.filter(x); dt=time.perf_counter()-t0
    print("host API 2^24 c64, strict_dtype=%s: %.1f ms -> %.1f MS/s (dtype %s)"%(strict, dt*1e3, n/dt/1e6, y.dtype))
