"""Developer aid: board power and shader clock while one workload runs back to back (is a kernel held by the power cap?):
   python tools/power_probe.py [option=value ...] [idle copy fir1024 fir1024f32 fir1024f64 fir1024c128 updn43 fir127 iir8 iir8cas iir8c64 iirlp8]
Samples `rocm-smi --showpower --showclocks --showperflevel` about twice a second from a second thread while the main thread keeps the queue full."""
import ctypes, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-dsp-comm_amd"))
import numpy as np
import bench
from sk_dsp_comm_amd import _ffi

def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return "rocm-smi failed: %s" % e
    keep = [l.strip() for l in out.splitlines() if ("Power" in l or "sclk" in l or "mclk" in l or "fclk" in l)]
    return " | ".join(keep)

def run(name, seconds=3.0):
    n = 1 << 26
    if name == "idle":
        step = None
    elif name == "copy":
        a = _ffi.DeviceArray(n, np.complex64).fill_noise(1); b = _ffi.DeviceArray(n, np.complex64)
        step = lambda: _ffi.load().skdsp_memcpy_d2d(ctypes.c_void_p(b.ptr), ctypes.c_void_p(a.ptr), 8 * n)
    elif name == "fir1024":
        k = _ffi.FirKernel(bench.firwin_lowpass(1024, 0.2), _ffi.C64)
        xd = _ffi.DeviceArray(n, np.complex64, headroom=1024).fill_noise(1); yd = _ffi.DeviceArray(n, np.complex64)
        step = lambda: k.filter_dev(xd, yd)
    elif name == "updn43":
        k = _ffi.FirKernel(bench.firwin_lowpass(512, 0.225), _ffi.C64)
        xd = _ffi.DeviceArray(n, np.complex64).fill_noise(1); yd = _ffi.DeviceArray(n * 4 // 3, np.complex64)
        step = lambda: k.updn_dev(xd, yd, 4, 3)
    elif name in ("fir1024f32", "fir1024f64", "fir1024c128"):
        dt = {"fir1024f32": np.float32, "fir1024f64": np.float64, "fir1024c128": np.complex128}[name]
        k = _ffi.FirKernel(bench.firwin_lowpass(1024, 0.2), _ffi.code_of(dt))
        xd = _ffi.DeviceArray(n, dt, headroom=1024).fill_noise(1); yd = _ffi.DeviceArray(n, dt)
        step = lambda: k.filter_dev(xd, yd)
    elif name == "fir127":
        k = _ffi.FirKernel(bench.firwin_lowpass(127, 0.2), _ffi.F32)
        xd = _ffi.DeviceArray(n, np.float32, headroom=1024).fill_noise(1); yd = _ffi.DeviceArray(n, np.float32)
        step = lambda: k.filter_dev(xd, yd)
    elif name in ("iir8", "iir8cas", "iir8c64", "iirlp8"):
        if name == "iirlp8":
            from scipy import signal
            sos = signal.butter(8, 0.9 / 12, output="sos")
        else:
            sos = np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]
        _ffi.set_option("iir_par", 0 if name == "iir8cas" else 1)
        dt = np.complex64 if name == "iir8c64" else np.float32
        k = _ffi.IirKernel(_ffi.code_of(dt), sos=sos)
        xd = _ffi.DeviceArray(n, dt).fill_noise(7); yd = _ffi.DeviceArray(n, dt)
        step = lambda: k.filter_dev(xd, yd)
    else:
        raise SystemExit("unknown workload " + name)
    samples, stop = [], threading.Event()
    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), smi()))
            stop.wait(0.4)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); calls = 0
    if step is None:
        time.sleep(seconds)
    else:
        for _ in range(50): step()
        _ffi.sync(); t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(200): step()
            _ffi.sync(); calls += 200
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    print("== %s: %d calls in %.2f s = %.4f ms per call (host clock, queue kept full)" % (name, calls, dt, 1e3 * dt / max(calls, 1)))
    for ts, s in samples: print("   t=%5.2f  %s" % (ts - t0, s))
    sys.stdout.flush()

_ffi.init(0)
opts = [a for a in sys.argv[1:] if "=" in a]      # library options for the whole run: name=value
for o in opts:
    _ffi.set_option(o.split("=")[0], int(o.split("=")[1]))
    print("option", o)
sys.argv = [a for a in sys.argv if "=" not in a]
for w in (sys.argv[1:] or ["idle", "fir1024", "updn43", "iir8", "fir127"]):
    run(w)
