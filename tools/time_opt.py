"""Time one bench workload for several values of one library option (same process, alternating):
python tools/time_opt.py <workload> <option> v1 v2 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi, sharding
wl, opt, vals = sys.argv[1], sys.argv[2], [int(v) for v in sys.argv[3:]]
tr = sharding.RcclTransport(0, 1, 0)
for rep in range(2):
    for v in vals:
        _ffi.set_option(opt, v)
        w = bench.make_workload(wl, 1 << 26, 0, 1, tr, _ffi, sharding)   # (fresh handle: plans are per handle)
        for _ in range(100): w.step()
        _ffi.sync(); _ffi.timer_start()
        for _ in range(300): w.step()
        ms = _ffi.timer_stop() / 300
        print("%s %s=%d: %.4f ms" % (wl, opt, v, ms), flush=True)
        bench.free_workload(w)
