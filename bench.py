#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X streaming-filter path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1: one process, one GPU.  N > 1: one rank per GPU over RCCL; either launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the ranks read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or, when no launcher set
WORLD_SIZE, bench.py spawns its own N ranks (same environment contract, 127.0.0.1
rendezvous) and forwards rank 0's line and the first non-zero exit code.

A "step" = one pass of multirate_FIR.filter (1024-tap lowpass, complex64) over each rank's
contiguous sample block, inputs already resident in HBM: halo exchange of the Ntaps-1 = 1023
preceding samples over RCCL (N > 1) beside the overlap-save kernel.  value = total samples all
ranks filtered / max-over-ranks time.

  default            weak scaling, 2^26 samples per GPU (the size the metric is quoted on)
  --scaling strong --total-log2n 30
                     BASELINE.json config 5 as written: 2^30 samples in total, 2^30 / N per GPU
                     (N = 1 runs all 2^30 on one GPU: 8 GiB in + 8 GiB out of its 288 GB)

The product path uses no PyTorch; the ranks rendezvous through a file
(sk_dsp_comm_amd.sharding.FileRendezvous) and barrier / max-reduce through RCCL.

Rank 0 prints ONE JSON line on stdout, the LAST thing it writes and kept under FINAL_LINE_MAX_BYTES
(final_line(); tests/test_host_cpu.py checks the bound on a recorded run and on the worst case): the
driver contract's fields + "roofline" + "cpu_baseline" (the C port of the reference's arithmetic) +
"cpu_baseline_scipy" (literally the reference's scipy.signal call) + "rows" (one compact entry per
other workload timed in the same process behind the headline: BASELINE.json configs 3, 4, the
127-tap shape, the SURVEY 8(a) rate rows) + for N > 1 the RCCL rank count, the halo form used, the
per-rank parity and the config-5 leg.  Everything longer (each other workload's full record, the
provenance of the PMC traffic, the device-copy reference) goes to stderr, one JSON line per workload
prefixed "bench.py detail: ", and to gpurun_out/bench_detail_n<N>.json -- never to stdout.
--workload updn43|iir8|fir127 runs one of the other configs as the main line instead.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "scikit-dsp-comm_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

# multi-process GPU work on this pool needs dmabuf IPC (already exported by the driver's environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def firwin_lowpass(ntaps, cutoff):
    """scipy.signal.firwin(ntaps, cutoff) (Hamming window, unit DC gain) restated so the
    bench does not need SciPy: fir_design_helper.firwin_lpf(n, fc) == firwin(n, 2*fc)."""
    m = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = cutoff * np.sinc(cutoff * m) * np.hamming(ntaps)
    return h / np.sum(h)


def elliptic_bpf_sos():
    """IIR_bpf(0.19,0.2,0.3,0.31,0.5,60,1.0,'ellip') from the golden fixture (8 biquads)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "g7_iir_sos.npz"))["sos8"]


# ------------------------------------------------------------------------------ per-rank parity of a sharded run
PARITY_TOL = 1e-6   # north_star: float32 / complex64 filtering within 1e-6 of the reference


def shard_parity(kind, coeffs, n_local, rank, get_x, get_y):
    """Checks what a rank just computed against the oracle (checker only), on the outputs that DEPEND on its left
    neighbour -- the first samples of the shard consume the Ntaps-1 halo (FIR) / the handed-over state (IIR) -- and on
    one interior window.  get_x(g0, count): `count` input samples of the GLOBAL signal from global index g0 (any rank's
    block: bench.py regenerates them from the counter-based noise, which is keyed by global index);
    get_y(i0, count): this rank's outputs.  -> (halo_err, interior_err), max-abs error / max-abs reference."""
    from oracle import oracle as orc
    g_start = rank * n_local
    m = min(2048, n_local)

    def rel(got, ref):
        return float(np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-300))

    if kind == "fir":
        b = coeffs
        hist = len(b) - 1
        h = min(hist, g_start)                                   # rank 0 starts from rest (lfilter's zero state)
        xs = get_x(g_start - h, h + m)
        halo_err = rel(get_y(0, m), orc.fir_filter(b, xs[h:], hist=xs[:h] if h else None))
        s0 = max(0, min(3 * 7168 - 100, n_local - m))
        h2 = min(hist, g_start + s0)
        xs = get_x(g_start + s0 - h2, h2 + m)
        return halo_err, rel(get_y(s0, m), orc.fir_filter(b, xs[h2:], hist=xs[:h2] if h2 else None))
    sos = coeffs
    lead = 1 << 15                                               # the cascade forgets its start within ~8000 samples (1e-18)
    h = min(lead, g_start)
    halo_err = rel(get_y(0, m), orc.sos_filter(sos, get_x(g_start - h, h + m))[h:])
    s0 = max(0, n_local // 2 - m)
    h2 = min(lead, g_start + s0)
    return halo_err, rel(get_y(s0, m), orc.sos_filter(sos, get_x(g_start + s0 - h2, h2 + m))[h2:])


HALO_FORMS = {0: "none yet", 1: "two launches (first step)", 2: "overlapped (probation passed)", 3: "two launches (probation failed)"}


def reduce_parity(tr, halo_err, interior_err, fallback_used, halo_state=0):
    """All ranks -> (max halo error, max interior error, per-rank fallback flags, ok, per-rank halo form).  The halo form is the
    state of the library's probation (dist.hip): a process's first sharded step runs two launches, its second the overlapped launch on
    probation, and only a passed probation makes the overlapped form the steady state."""
    tab = tr.allgather_state(np.array([halo_err, interior_err, float(fallback_used), float(halo_state)]))
    hmax, imax = float(np.max(tab[:, 0])), float(np.max(tab[:, 1]))
    ok = bool(np.isfinite(hmax) and np.isfinite(imax) and hmax <= PARITY_TOL and imax <= PARITY_TOL)
    return hmax, imax, [int(v) for v in tab[:, 2]], ok, [int(v) for v in tab[:, 3]]


# ------------------------------------------------------------------------------ BASELINE config 5 inside an N > 1 run
CONFIG5_TOTAL_LOG2N = 30


def config5_n1_reference():
    """The N = 1 figure of BASELINE config 5 (all 2^30 samples on ONE GPU): the newest committed
    profiles/rNN/bench_fir1024_2p30_one_gpu.json (`python bench.py --scaling strong --total-log2n 30`).  That line records the
    hashes of the kernel sources it ran; when they differ from the sources on disk the figure is withheld, like roofline.traffic."""
    prof = os.path.join(ROOT, "profiles")
    best = None
    for d in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        f = os.path.join(prof, d, "bench_fir1024_2p30_one_gpu.json")
        if os.path.exists(f):
            best = f
    if best is None:
        return {"ms": None, "why": "no committed profiles/rNN/bench_fir1024_2p30_one_gpu.json"}
    try:
        j = json.loads(open(best).read().strip().splitlines()[-1])
        ref = {"profile": os.path.relpath(best, ROOT), "ms": float(j["ms_per_step"]), "value": float(j["value"]),
               "kernel_source_sha256": j.get("kernel_source_sha256")}
        if int(j["config"]["total_samples"]) != 1 << CONFIG5_TOTAL_LOG2N or int(j["n_gpus"]) != 1:
            return {"ms": None, "why": "%s is not a 1-GPU run over 2^%d samples" % (ref["profile"], CONFIG5_TOTAL_LOG2N)}
        if j.get("kernel_source_sha256") != source_hashes("fir1024"):
            ref["stale"] = "the kernel sources changed after this N = 1 line was recorded: speedup withheld"
            ref["ms"] = None
        return ref
    except Exception as e:
        return {"ms": None, "why": "%s: %s" % (type(e).__name__, e)}


def config5_summary(total, world, step_ms, kernel_ms, parity, n1):
    """BASELINE config 5 as one record: `total` samples split over `world` GPUs (strong scaling), whole-job MSamples/s, the
    per-GPU fraction of the HBM roofline (16 B per sample over the slowest rank's step), and the speed-up over the committed
    N = 1 run of the same 2^30 samples (north_star: >= 6x at 8 GPUs)."""
    out = {"what": "BASELINE config 5: 1024-tap complex64 FIR, 2^%d samples sharded by contiguous sample block over %d GPU(s), "
                   "%d-sample RCCL halo r -> r+1 beside the interior tiles (strong scaling)" % (total.bit_length() - 1, world, 1023),
           "total_samples": total, "samples_per_gpu": total // world, "n_gpus": world, "ms": step_ms, "kernel_ms_max": kernel_ms,
           "value": total / (step_ms * 1e-3) / 1e6, "unit": "MSamples/s",
           "frac_per_gpu": 16.0 * (total // world) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "n1_reference": n1, "speedup_vs_n1": (n1["ms"] / step_ms) if n1.get("ms") else None}
    if parity is not None:
        out["parity_halo_max_err"], out["parity_interior_max_err"], out["halo_fallback_used"], out["parity_ok"], out["halo_state"] = parity
    return out


def config5_leg(args, rank, world, tr, _ffi, sharding):
    """After the weak-scaling steps of an N > 1 run: the same filter over 2^30 samples in total (2^30 / N per rank), timed and
    verified exactly like the main line (barrier + sync on both sides, max over ranks; every rank checks the outputs that consumed
    its neighbour's halo).  -> (record on rank 0 / None elsewhere, parity ok)."""
    total = 1 << CONFIG5_TOTAL_LOG2N
    if total % world:
        return ({"skipped": "%d GPUs do not divide 2^%d samples" % (world, CONFIG5_TOTAL_LOG2N)} if rank == 0 else None), True
    n5 = total // world
    w5, err = None, None
    try:
        w5 = make_workload("fir1024", n5, rank, world, tr, _ffi, sharding)
    except Exception as e:      # (e.g. out of memory on one rank: every rank must learn of it BEFORE the collectives below)
        err = "%s: %s" % (type(e).__name__, e)
    if tr.allreduce_max(1.0 if err else 0.0) > 0:
        if w5 is not None:
            free_workload(w5)
        return ({"error": err or "another rank could not allocate its 2^%d / %d samples" % (CONFIG5_TOTAL_LOG2N, world)} if rank == 0 else None), False
    try:
        K5 = max(10, min(args.steps, 50))
        elapsed, ev_ms, _ = timed_steps(w5, K5, 10, 0.2, tr, _ffi)
        kind, coeffs, seed = w5.shard

        def get_x(g0, count):
            t = _ffi.DeviceArray(count, w5.dtype).fill_noise(seed, first_index=g0)
            try:
                return t.to_host()
            finally:
                t.free()
        herr, ierr = shard_parity(kind, coeffs, n5, rank, get_x, lambda i0, c: w5.yd.to_host(i0, c))
        parity = reduce_parity(tr, herr, ierr, _ffi.get_option("shard_two_launches"), _ffi.get_option("shard_halo_state"))
    finally:
        free_workload(w5)
    rec = None
    if rank == 0:
        rec = config5_summary(total, world, elapsed * 1e3 / K5, ev_ms / K5, parity, config5_n1_reference())
        rec["steps"] = K5
    return rec, parity[3]


# ------------------------------------------------------------------------------ launcher
def self_launch(args):
    """No launcher set WORLD_SIZE: spawn one rank per GPU ourselves (torchrun's environment contract)."""
    n = args.gpus
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, deadline = 0, time.time() + args.launch_timeout
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:          # one rank failed: the others would wait for it forever
                    q.terminate()
        if time.time() > deadline:
            rc = rc or 124
            for q in live:
                q.kill()
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


# ------------------------------------------------------------------------------ workloads
class Workload:
    """step() = one pass; units = input samples per rank per pass."""


def make_workload(name, n, rank, world, tr, _ffi, sharding):
    w = Workload()
    w.name, w.n, w.units, w.compute, w.check, w.taps, w.shard = name, n, n, None, None, None, None
    lg = "2^%d" % (n.bit_length() - 1) if n & (n - 1) == 0 else str(n)
    if name == "fir1024":
        b = firwin_lowpass(1024, 0.2)
        w.taps, w.dtype, w.arith = b, np.complex64, "c64"
        fir = sharding.ShardedFIR(b, tr, dtype=w.dtype)
        w.xd = fir.new_shard_buffer(n).fill_noise(2026, first_index=rank * n)
        w.yd = _ffi.DeviceArray(n, w.dtype)
        w.step = lambda: fir.filter_local_dev(w.xd, w.yd, n)
        w.shard = ("fir", b, 2026)                               # per-rank halo / interior parity (shard_parity)
        w.alg_bytes = 16.0 * n                                   # 8 B in + 8 B out per sample
        w.kern = "ols_tile_kernel"
        w.wl = "multirate_FIR.filter: 1024-tap lowpass, complex64, %s samples per GPU, FFT overlap-save" % lg
        w.metric = "complex64 MSamples/s (FIR-1024 tap, %s samples%s)" % (lg, " per GPU" if world > 1 else "")

        def check():
            from oracle import oracle as orc
            s0, wn = 3 * 7168 - 100, 2048
            ref = orc.fir_filter(b, w.xd.to_host(s0 - 1023, wn + 1023))[1023:]
            got = w.yd.to_host(s0, wn)
            return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
        w.check = check
    elif name == "fir127":
        b = firwin_lowpass(127, 0.2)
        w.taps, w.dtype, w.arith = b, np.float32, "f32"
        k = _ffi.FirKernel(b, _ffi.F32)
        w.xd = _ffi.DeviceArray(n, w.dtype).fill_noise(2026)
        w.yd = _ffi.DeviceArray(n, w.dtype)
        w.step = lambda: k.filter_dev(w.xd, w.yd)
        w.alg_bytes = 8.0 * n
        w.kern = "fir_bx_kernel"
        w.wl = ("multirate_FIR.filter: 127-tap lowpass, float32, %s samples, direct form on the fp16 matrix pipe "
                "(two fp16 pieces per operand, three products per tap = float32 accuracy)" % lg)
        w.metric = "float32 MSamples/s (FIR-127 tap, %s samples)" % lg
    elif name == "updn43":
        b = firwin_lowpass(512, 0.225)
        w.taps, w.dtype, w.arith = b, np.complex64, "c64"
        k = _ffi.FirKernel(b, _ffi.C64)
        w.xd = _ffi.DeviceArray(n, w.dtype).fill_noise(2026)
        n_out = (n * 4) // 3
        w.yd = _ffi.DeviceArray(n_out, w.dtype)
        w.step = lambda: k.updn_dev(w.xd, w.yd, 4, 3)
        w.alg_bytes = 8.0 * n + 8.0 * n_out                      # 18.67 B per input sample
        # float32 arithmetic carried by 3 fp16 products per multiply: against the FP32 matrix / vector peak the useful
        # flops may exceed 100 % -- that is the point of the split
        w.compute = ("useful f32 flops vs the FP32 matrix-pipe peak (computed as 3 fp16 MFMA products per multiply)",
                     157.3, 4.0 * 512 / 4 * n_out)               # 4*Ntaps/L flop per c64 output
        w.kern = "fir_bx_kernel"
        w.wl = ("downsample(multirate_FIR.up(x,4),3): 512-tap prototype, complex64, %s input samples, fused polyphase "
                "(Toeplitz product on the fp16 matrix pipe, two fp16 pieces per operand = float32 accuracy)" % lg)
        w.metric = "complex64 input MSamples/s (polyphase L=4/M=3, 512 taps, %s samples)" % lg
    elif name == "fir1024c128":
        b = firwin_lowpass(1024, 0.2)
        w.taps, w.dtype, w.arith = b, np.complex128, "c128"
        k = _ffi.FirKernel(b, _ffi.C128)
        w.xd = _ffi.DeviceArray(n, w.dtype).fill_noise(2026)
        w.yd = _ffi.DeviceArray(n, w.dtype)
        w.step = lambda: k.filter_dev(w.xd, w.yd)
        w.alg_bytes = 32.0 * n
        w.kern = "ols64_tile_kernel (float64 overlap-save, 4096-point tiles)"
        w.wl = "multirate_FIR.filter: 1024-tap lowpass, complex128 (the reference's own arithmetic), %s samples" % lg
        w.metric = "complex128 MSamples/s (FIR-1024 tap, %s samples)" % lg
    elif name in ("iir8", "iir8tp", "iir8cas", "iirlp8", "iir8c64"):
        if name == "iirlp8":   # rate_change(12)'s own design (multirate_helper.py:62): butter(8, 0.9 / 12), as biquads
            sos = _ffi.tf2sos(*_butter8_rate_change12())
        else:
            sos = elliptic_bpf_sos()
        w.sos, w.dtype, w.arith = sos, np.float32, "f32 I/O, f64 state"
        if name == "iir8c64":   # the same cascade on a complex baseband signal: re and im streams interleaved on alternate lanes
            w.dtype, w.arith = np.complex64, "c64 I/O, f64 state"
        w.xd = _ffi.DeviceArray(n, w.dtype).fill_noise(2026, first_index=rank * n)
        w.yd = _ffi.DeviceArray(n, w.dtype)
        if world == 1:
            k = _ffi.IirKernel(_ffi.code_of(w.dtype), sos=sos)
            if name == "iir8tp":   # config 4 through K1 + carries + K3 (what round 1 ran), for comparison
                def step():
                    with _ffi.option("iir_par", 0), _ffi.option("iir_two_pass", 1):
                        k.filter_dev(w.xd, w.yd)
                w.step = step
            elif name == "iir8cas":   # config 4 through the cascade-form single-pass scan (what round 2 ran), for comparison
                def step():
                    with _ffi.option("iir_par", 0):
                        k.filter_dev(w.xd, w.yd)
                w.step = step
            else:
                w.step = lambda: k.filter_dev(w.xd, w.yd)
        else:  # contiguous sample blocks, exact state hand-off rank r -> r+1 (16 doubles per hop)
            iir = sharding.ShardedIIR(sos, tr, dtype=w.dtype)
            w.step = lambda: iir.filter_local_dev(w.xd, w.yd, n)
        if name == "iir8":
            w.shard = ("iir", sos, 2026)
        cplx = 2 if name == "iir8c64" else 1
        w.alg_bytes = 8.0 * n * cplx
        w.compute = ("FP64 vector (v_fma_f64)", 78.6, 72.0 * n * cplx)   # 9 flop per biquad per (real) sample
        if name in ("iir8", "iir8c64"):
            # what actually bounds this kernel: v_fma_f64 and v_mfma_f64 share ONE datapath on this chip
            # (tools/ubench_dp_pipes.hip), and a real sample costs 33 (recurrence + output taps) + 2 (conversions) + 16 (from-rest
            # end states on the matrix pipe) + ~3 (scan, correction) issue slots of 64 lanes x 4 cycles
            w.dp_slots = 54.0 * n * cplx
        w.kern = {"iir8": "iir_par_kernel (parallel-form single-pass scan, one segment per wave)",
                  "iir8cas": "iir_fused_kernel (cascade-form single-pass scan, chunk scan on the matrix pipe; forced)",
                  "iir8tp": "iir_k1r_kernel + iir_carry_kernel + iir_chunk_kernel (two-pass scan, forced)",
                  "iirlp8": "iir_par_kernel (parallel-form single-pass scan)",
                  "iir8c64": "iir_par_kernel<complex> (parallel-form single-pass scan, re / im on alternate lanes)"}[name]
        what = "order-8 Butterworth lowpass of rate_change(12), 4 biquads" if name == "iirlp8" else "8-biquad elliptic bandpass"
        w.wl = "multirate_IIR.filter: %s, %s, %s samples, exact affine scan%s" % (
            what, "complex64" if name == "iir8c64" else "float32", lg, " (two-pass scan forced)" if name == "iir8tp" else " (cascade-form single pass forced)" if name == "iir8cas" else "")
        w.metric = "%s MSamples/s (%s, %s samples)" % ("complex64" if name == "iir8c64" else "float32", "SOS IIR, " + what, lg)
        if name == "iirlp8":
            w.compute = ("FP64 vector (v_fma_f64)", 78.6, 36.0 * n)
    elif name in RATE_WORKLOADS:
        make_rate_workload(w, name, n, lg, _ffi)
    else:
        raise ValueError(name)
    if w.check is None and world == 1:
        w.check = lambda: head_check(w)
    return w


def head_check(w, m=6000):
    """The head of what a single-GPU workload just computed against the oracle (checker only): max-abs error / max-abs reference."""
    from oracle import oracle as orc
    x = w.xd.to_host(0, m)
    if w.name == "updn43":
        ref = orc.downsample(orc.fir_up(w.taps, x, 4), 3)
    elif getattr(w, "sos", None) is not None:
        ref = orc.sos_filter(w.sos, x)
    else:
        ref = orc.fir_filter(w.taps, x)
    got = w.yd.to_host(0, ref.size)
    return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))


# The .up / .dn / resampling rows of SURVEY.md 8(a), each at the reference's defaults, sized by the HIGH rate (n samples on the fast side):
# name -> (reference call, file:line under /root/reference/src/sk_dsp_comm)
RATE_WORKLOADS = {
    "upsample4": ("sigsys.upsample(x, 4), complex64", "sigsys.py:3031-3053"),
    "downsample3": ("sigsys.downsample(x, 3), complex64", "sigsys.py:3056-3083"),
    "firup12": ("multirate_FIR(512-tap lowpass).up(x) at the default L_change = 12, complex64", "multirate_helper.py:112-118"),
    "firdn12": ("multirate_FIR(512-tap lowpass).dn(x) at the default M_change = 12, complex64", "multirate_helper.py:121-127"),
    "firup4": ("multirate_FIR(1024-tap lowpass).up(x, 4), complex64 (overlap-save on tiles of the output: the zero-stuffed tile's spectrum from its non-zero columns)", "multirate_helper.py:112-118"),
    "firdn4": ("multirate_FIR(1024-tap lowpass).dn(x, 4), complex64 (overlap-save, decimating inverse transform: the spectrum folded 4-fold)", "multirate_helper.py:121-127"),
    "rcup12": ("rate_change(12).up(x): order-8 Butterworth, float32", "multirate_helper.py:69-75"),
    "rcdn12": ("rate_change(12).dn(x): order-8 Butterworth, float32", "multirate_helper.py:77-83"),
    "iirup2": ("multirate_IIR(8-biquad elliptic bandpass).up(x, 2), float32", "multirate_helper.py:177-184"),
    "iirdn3": ("multirate_IIR(8-biquad elliptic bandpass).dn(x, 3), float32", "multirate_helper.py:186-192"),
}


def make_rate_workload(w, name, n, lg, _ffi):
    """n = samples at the HIGH rate (outputs of an interpolator, inputs of a decimator).  Algorithmic bytes as SURVEY.md 8(d) counts
    them: every input sample read once, every output sample written once."""
    what, ref = RATE_WORKLOADS[name]
    up = name in ("upsample4", "firup12", "firup4", "rcup12", "iirup2")
    R = {"upsample4": 4, "downsample3": 3, "firup12": 12, "firdn12": 12, "firup4": 4, "firdn4": 4, "rcup12": 12, "rcdn12": 12, "iirup2": 2, "iirdn3": 3}[name]
    cplx = name in ("upsample4", "downsample3", "firup12", "firdn12", "firup4", "firdn4")
    w.dtype, w.arith = (np.complex64, "c64") if cplx else (np.float32, "f32 I/O, f64 state")
    esz = 8 if cplx else 4
    n_in = n // R if up else n
    n_out = n_in * R if up else n_in // R
    w.units = n_in
    w.xd = _ffi.DeviceArray(n_in, w.dtype).fill_noise(2026)
    w.yd = _ffi.DeviceArray(n_out, w.dtype)
    w.alg_bytes = float(esz) * (n_in + n_out)
    code = _ffi.code_of(w.dtype)
    L = _ffi.load()
    import ctypes
    xp, yp = ctypes.c_void_p(w.xd.ptr), ctypes.c_void_p(w.yd.ptr)
    if name == "upsample4":
        w.step = lambda: _ffi.check(L.skdsp_upsample_dev(xp, n_in, R, code, ctypes.c_double(1.0), yp))
        w.kern = "upsample_kernel (resample.hip)"
    elif name == "downsample3":
        w.step = lambda: _ffi.check(L.skdsp_downsample_dev(xp, n_in, R, 0, code, yp))
        w.kern = "downsample_tile_kernel (resample.hip)"
    elif name.startswith("fir"):
        b = firwin_lowpass(1024, 0.2 / 4) if name in ("firup4", "firdn4") else firwin_lowpass(512, 0.9 / 12)
        w.taps = b
        k = _ffi.FirKernel(b, _ffi.C64)
        w.step = (lambda: k.up_dev(w.xd, w.yd, R)) if up else (lambda: k.dn_dev(w.xd, w.yd, R))
        w.kern = "AUTO dispatch of skdsp_fir_%s_dev (%d taps, %d per phase)" % ("up" if up else "dn", len(b), -(-len(b) // R))
    else:
        sos = _ffi.tf2sos(*_butter8_rate_change12()) if name.startswith("rc") else elliptic_bpf_sos()
        w.sos = sos
        k = _ffi.IirKernel(code, sos=sos)
        w.step = (lambda: k.up_dev(w.xd, w.yd, R)) if up else (lambda: k.dn_dev(w.xd, w.yd, R))
        w.kern = "iir_par_kernel (parallel-form single-pass scan; %s)" % ("zero-stuffing fused into the staging" if up else "kept outputs gathered on chip")
    w.rate = {"factor": R, "direction": "up" if up else "dn", "n_in": n_in, "n_out": n_out}
    w.wl = "%s [%s]: %s samples at the high rate (%d in, %d out)" % (what, ref, lg, n_in, n_out)
    w.metric = "%s input MSamples/s (%s, %s high-rate samples)" % ("complex64" if cplx else "float32", name, lg)

    def check():   # the head of the result against the oracle (checker only; the IIR paths start from rest like the reference)
        from oracle import oracle as orc
        m = 6000
        x = w.xd.to_host(0, m)
        if name == "upsample4":
            return float(np.max(np.abs(w.yd.to_host(0, m * R) - orc.upsample(x, R).astype(w.dtype))))
        if name == "downsample3":
            return float(np.max(np.abs(w.yd.to_host(0, m // R) - x[: (m // R) * R: R])))
        if name.startswith("fir"):
            ref = orc.fir_up(w.taps, x, R) if up else orc.downsample(orc.fir_filter(w.taps, x), R)
        else:
            ref = orc.sos_filter(w.sos, R * orc.upsample(x, R)) if up else orc.downsample(orc.sos_filter(w.sos, x), R)
        got = w.yd.to_host(0, ref.size)
        return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    w.check = check


def _butter8_rate_change12():
    """(b, a) of scipy.signal.butter(8, 0.9 / 12) -- the design rate_change(12) makes (multirate_helper.py:58-62) --
    restated (bilinear transform of the analog Butterworth prototype) so that the bench does not need SciPy."""
    n, wn = 8, 0.9 / 12
    warped = 2.0 * 2.0 * np.tan(np.pi * wn / 2.0)                      # fs = 2
    m = np.arange(-n + 1, n, 2)
    p = -np.exp(1j * np.pi * m / (2 * n)) * warped                     # analog poles, cutoff warped
    k = warped ** n
    pz = (4.0 + p) / (4.0 - p)                                         # bilinear, fs = 2
    kz = k * np.real(1.0 / np.prod(4.0 - p))
    b = kz * np.poly(-np.ones(n))
    a = np.real(np.poly(pz))
    return np.real(b), a


def free_workload(w):
    for a in ("xd", "yd"):
        d = getattr(w, a, None)
        if d is not None:
            d.free()


def timed_steps(w, K, W, settle_s, tr, _ffi):
    """Settle the clock, W warm-up passes, then K timed passes bracketed by barrier + sync.
    -> (wall seconds max over ranks, HIP-event ms max over ranks, this rank's (wall, event ms))."""
    # The chip idles at ~160 MHz; the first ~50 launches after idle run on a ramping clock (0.28 ms for the first
    # 50-launch window of the headline kernel, 0.236 ms from the second window on, flat for as long as the launches
    # continue: profiles/r01/README.md).  Steady state is what a streaming job sees, so the clock is settled first,
    # untimed, whatever W is.  (The number of settle passes is agreed between the ranks: every pass of a sharded
    # workload contains a send/recv pair, so a per-rank time-based loop would deadlock.)
    t_settle = time.perf_counter()
    for _ in range(10):
        try:
            w.step()
        except _ffi.SkdspError as e:  # (the sharded FIR switches itself to its two-launch form if the RCCL receive
            if "two-launch" not in str(e):  # cannot run beside the persistent launch here; the step is simply repeated)
                raise
            print("bench.py: %s" % e, file=sys.stderr)
            w.step()   # (the refused call enqueued nothing: repeat it so that every rank has made the same number of exchanges)
    try:
        _ffi.sync()
    except _ffi.SkdspError as e:   # a step of the settle phase gave up waiting for its halo inside the launch: reported at
        if "two-launch" not in str(e):   # this sync, the library has switched to the two-launch form; the timed steps use it
            raise
        print("bench.py: %s" % e, file=sys.stderr)
    per_pass = max((time.perf_counter() - t_settle) / 10, 1e-6)
    n_settle = int(tr.allreduce_max(float(min(20000, int(settle_s / per_pass) + 1)))) if settle_s > 0 else 0
    for _ in range(n_settle):
        w.step()
    _ffi.sync()
    for _ in range(W):
        w.step()
    _ffi.sync()
    tr.barrier()
    _ffi.timer_start()
    t0 = time.perf_counter()
    for _ in range(K):
        w.step()
    ev_ms = _ffi.timer_stop()  # HIP events on the stream the kernels run on (synchronises)
    _ffi.sync()
    tr.barrier()
    t1 = time.perf_counter()
    return tr.allreduce_max(t1 - t0), tr.allreduce_max(ev_ms), (t1 - t0, ev_ms)


def _smi_sample():
    """(board power W, power cap W, shader clock MHz) from one rocm-smi call; None where it could not be read."""
    import re, subprocess
    try:
        txt = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks"], capture_output=True, text=True, timeout=30).stdout
    except Exception:
        return None, None, None
    def grab(pat):
        m = re.search(pat, txt)
        return float(m.group(1)) if m else None
    return (grab(r"Current Socket Graphics Package Power \(W\):\s*([0-9.]+)") or grab(r"Average Graphics Package Power \(W\):\s*([0-9.]+)"),
            grab(r"Max Graphics Package Power \(W\):\s*([0-9.]+)"), grab(r"sclk clock level:[^(]*\(([0-9.]+)Mhz\)"))


def board_state_of(w, seconds, _ffi):
    """Board power and shader clock (rocm-smi, a second thread) while the same step runs back to back for `seconds`,
    right after the timed region.  The streaming kernels here run AT the board power cap: the shader clock, and with it the
    time per step, is what the firmware leaves under that cap, which is why instruction-level changes that do not save
    energy do not move these numbers."""
    import threading
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(_smi_sample())
            stop.wait(0.2)
    th = threading.Thread(target=sampler)
    t0 = time.perf_counter()
    for _ in range(50):
        w.step()
    _ffi.sync()
    th.start()
    while time.perf_counter() - t0 < seconds:
        for _ in range(100):
            w.step()
        _ffi.sync()
    stop.set()
    th.join()
    pw = sorted(v[0] for v in samples if v[0] is not None)
    ck = sorted(v[2] for v in samples if v[2] is not None)
    cap = [v[1] for v in samples if v[1] is not None]
    if not pw:
        return {"skipped": "rocm-smi gave no reading on this box"}
    return {"power_w": pw[len(pw) // 2], "power_cap_w": cap[0] if cap else None, "sclk_mhz": ck[len(ck) // 2] if ck else None,
            "sclk_max_mhz": 2400, "samples": len(pw),
            "what": "median rocm-smi reading while this step runs back to back for %.1f s after the timed region" % seconds}


def device_copy_of(w, _ffi, reps=40):
    """What this board streams when it does nothing else: hipMemcpyAsync device-to-device of as many bytes as the workload reads,
    counted like the roofline counts (bytes read + bytes written), timed with HIP events on the same stream."""
    import ctypes
    nbytes = int(min(w.xd.n * w.xd.dtype.itemsize, w.yd.n * w.yd.dtype.itemsize))
    cp = _ffi.load().skdsp_memcpy_d2d
    dst, src = ctypes.c_void_p(w.yd.ptr), ctypes.c_void_p(w.xd.ptr)
    for _ in range(10):
        _ffi.check(cp(dst, src, nbytes))
    _ffi.sync()
    _ffi.timer_start()
    for _ in range(reps):
        _ffi.check(cp(dst, src, nbytes))
    ms = _ffi.timer_stop() / reps
    out = {"GBps": 2 * nbytes / ms / 1e6, "ms": ms, "bytes_copied": nbytes, "what": "hipMemcpyAsync device-to-device, read + written bytes"}
    # the library's own stride-1 copy (sigsys.downsample(x, 1): one 16-byte load and store per lane) over the same bytes: a second
    # reference, because hipMemcpyAsync is not the fastest copy this board does
    try:
        dn = _ffi.load().skdsp_downsample_dev
        n_el = nbytes // w.xd.dtype.itemsize
        for _ in range(10):
            _ffi.check(dn(src, n_el, 1, 0, w.xd.code, dst))
        _ffi.sync()
        _ffi.timer_start()
        for _ in range(reps):
            _ffi.check(dn(src, n_el, 1, 0, w.xd.code, dst))
        ms2 = _ffi.timer_stop() / reps
        out["own_copy_GBps"] = 2 * nbytes / ms2 / 1e6
        out["own_copy_what"] = "skdsp_downsample_dev(x, n, M=1): this library's stride-1 copy kernel over the same bytes"
    except Exception as e:
        out["own_copy_GBps"] = None
        out["own_copy_what"] = "%s: %s" % (type(e).__name__, e)
    return out


def roofline_of(w, ev_ms, K, log2n_for_traffic):
    t_kernel = ev_ms * 1e-3 / K  # average launch (+ halo) duration from HIP events
    achieved = w.alg_bytes / t_kernel / 1e9
    traffic, traffic_source = measured_traffic(w.name, log2n_for_traffic)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": traffic_source, "kernel": w.kern, "kernel_ms": t_kernel * 1e3,
            "algorithmic_bytes_per_launch": w.alg_bytes}


def compute_of(w, ev_ms, K):
    if w.compute is None:
        return None
    # these workloads are also priced against arithmetic: useful flops of the reference
    # formulation against the peak of the unit that executes them
    tf = w.compute[2] / (ev_ms * 1e-3 / K) / 1e12
    out = {"unit": "TFLOP/s", "what": w.compute[0], "useful_flop_per_step": w.compute[2], "achieved": tf, "peak": w.compute[1]}
    if tf <= w.compute[1]:
        out["frac"] = tf / w.compute[1]
    else:   # the kernel does not execute the formulation these flops were counted in: a fraction above 1 would mean nothing
        out["frac"] = None
        out["note"] = "useful flops of the reference formulation exceed the peak of the unit named: the kernel runs a cheaper algorithm"
    return out


def compact_row(ms, frac, traffic, alg_bytes, err, board):
    r = {"ms": round(ms, 4), "frac": round(frac, 3), "tr": round(traffic / alg_bytes, 3) if traffic else None,
         "err": float("%.1e" % err) if err is not None else None}
    if board and board.get("power_w") is not None:
        r["W"], r["MHz"] = int(board["power_w"]), int(board["sclk_mhz"] or 0)
    return r


def dp_pipe_of(w, ev_ms, K, compute_units):
    """FP64-pipe utilisation of the parallel-form IIR kernel: executed FP64 issue slots (vector + matrix: one datapath) against
    what the chip can issue at its nominal 2.4 GHz (4 SIMDs x 16 lanes per CU and cycle; the part sustains 1.9-2.1 GHz here)."""
    slots = getattr(w, "dp_slots", None)
    if not slots:
        return None
    t = ev_ms * 1e-3 / K
    peak = compute_units * 4 * 16 * 2.4e9        # lane-slots per second
    return {"what": "FP64 issue slots per step (v_fma_f64 and v_mfma_f64 share one datapath: tools/ubench_dp_pipes.hip)",
            "slots_per_step": slots, "slots_per_real_sample": 54.0, "achieved_per_s": slots / t, "peak_per_s_at_2.4GHz": peak,
            "frac": slots / t / peak}


# ------------------------------------------------------------------------------ what rank 0 prints
FINAL_LINE_MAX_BYTES = 8000   # the driver parses the LAST stdout line out of a bounded tail (a 22.9 KB line was not kept: BENCH_r05.json)
ROWS_WHAT = "per workload (2^26 samples): kernel ms, fraction of 8 TB/s, PMC traffic / algorithmic bytes, head error vs the oracle, board W, shader MHz"


def _clip(v, n):
    v = str(v)
    return v if len(v) <= n else v[: n - 3] + "..."


def _sig(v, digits=6):
    if isinstance(v, float) and np.isfinite(v) and v != int(v):     # (whole numbers -- byte and sample counts -- stay exact)
        return float("%.*g" % (digits, v))
    return v


def _pick(d, keys, clip=None):
    out = {}
    for k in keys:
        if isinstance(d, dict) and k in d:
            v = d[k]
            out[k] = _clip(v, clip) if (clip and isinstance(v, str)) else _sig(v) if isinstance(v, float) else v
    return out


def final_line(out):
    """The full record of a run -> the ONE line that goes to stdout: the driver contract's fields, `roofline`, `cpu_baseline`,
    `cpu_baseline_scipy`, the compact `rows` table and, for N > 1, what a scaling record needs (RCCL rank count, which halo form ran,
    per-rank parity, per-rank times, the config-5 leg with its N = 1 cross-check) -- at most FINAL_LINE_MAX_BYTES bytes whatever the
    run recorded.  Pure (no device, no clock): tests/test_host_cpu.py runs it on a recorded line and on the worst case."""
    fl = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                  "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    fl["config"] = _pick(cfg, ("workload", "samples_per_gpu", "total_samples", "clock_settle_s", "sharding", "n_ranks_rccl",
                               "device", "compute_units", "logical_device"), clip=200)
    r = out.get("roofline") or {}
    fl["roofline"] = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                               "algorithmic_bytes_per_launch", "frac_of_device_copy", "frac_of_fastest_copy"), clip=120)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):          # the members the contract names are always there
        fl["roofline"].setdefault(k, None)
    src = r.get("traffic_source") or {}
    if src:
        fl["roofline"]["traffic_profile"] = _clip(src.get("stale") or src.get("profile"), 100)
    for k in ("cpu_baseline", "cpu_baseline_scipy"):
        if k in out:
            fl[k] = _pick(out[k], ("value", "unit", "cores", "kind", "sample", "skipped", "cores_available"), clip=230) if out[k] else None
    if "kernel_source_sha256" in out:   # (which kernel sources this line ran: what config5_n1_reference() checks a committed N = 1 line against)
        fl["kernel_source_sha256"] = out["kernel_source_sha256"]
    if "parity_spot_check_max_err" in out:
        fl["parity_spot_check_max_err"] = _sig(out["parity_spot_check_max_err"], 3)
    if isinstance(out.get("board"), dict):
        fl["board"] = _pick(out["board"], ("power_w", "power_cap_w", "sclk_mhz", "skipped"), clip=80)
    if isinstance(out.get("device_copy"), dict):
        fl["device_copy"] = _pick(out["device_copy"], ("GBps", "own_copy_GBps", "skipped"), clip=80)
    for k in ("compute", "fp64_pipe"):
        if isinstance(out.get(k), dict):
            fl[k] = _pick(out[k], ("unit", "achieved", "peak", "frac", "slots_per_real_sample"))
    # ---- N > 1 (a future SCALE record must parse at every N): which communicator, which halo form, parity of every rank
    fl["n_ranks_rccl"] = cfg.get("n_ranks_rccl")
    for k in ("halo_fallback_used", "halo_state", "halo_form", "parity_halo_max_err", "parity_interior_max_err", "parity_ok"):
        if k in out:
            fl[k] = _sig(out[k], 3) if isinstance(out[k], float) else out[k]
    if isinstance(out.get("per_rank"), dict):
        fl["per_rank"] = {k: [round(float(v), 4) for v in vs][:64] for k, vs in out["per_rank"].items()}
    c5 = out.get("config5")
    if isinstance(c5, dict):
        f5 = _pick(c5, ("total_samples", "samples_per_gpu", "n_gpus", "steps", "ms", "kernel_ms_max", "value", "unit", "frac_per_gpu",
                        "speedup_vs_n1", "parity_halo_max_err", "parity_interior_max_err", "halo_fallback_used", "halo_state", "parity_ok",
                        "error", "skipped"), clip=160)
        n1 = c5.get("n1_reference") or {}
        f5["n1_cross_check"] = _pick(n1, ("profile", "ms", "value", "stale", "why"), clip=120)   # the committed 1-GPU run of the same 2^30 samples
        fl["config5"] = f5
    if "rows" in out:
        fl["rows_what"] = ROWS_WHAT
        fl["rows"] = out["rows"]
    if out.get("detail"):
        fl["detail"] = out["detail"]
    # ---- the bound, whatever was recorded: shed the least important members first
    for drop in ("rows_what", "device_copy", "compute", "fp64_pipe", "per_rank", "detail", "kernel_source_sha256"):
        if len(json.dumps(fl)) <= FINAL_LINE_MAX_BYTES:
            break
        fl.pop(drop, None)
    if len(json.dumps(fl)) > FINAL_LINE_MAX_BYTES and "rows" in fl:   # rows as [ms, frac] pairs
        fl["rows"] = {k: ([v.get("ms"), v.get("frac")] if "ms" in v else "error") for k, v in list(fl["rows"].items())[:40]}
    for drop in ("rows", "config5", "cpu_baseline_scipy"):
        if len(json.dumps(fl)) <= FINAL_LINE_MAX_BYTES:
            break
        fl.pop(drop, None)
    assert len(json.dumps(fl)) <= FINAL_LINE_MAX_BYTES, len(json.dumps(fl))
    return fl


def emit(out):
    """Rank 0, once: the long records to stderr and to a side file, then the one line on stdout."""
    n = out.get("n_gpus", 1)
    for name, o in (out.get("other_configs") or {}).items():
        print("bench.py detail: " + json.dumps(dict(o, name=name)), file=sys.stderr)
    rest = {k: v for k, v in out.items() if k not in ("other_configs", "rows", "rows_what")}
    print("bench.py detail: " + json.dumps(dict(rest, name="headline")), file=sys.stderr)
    sys.stderr.flush()
    try:   # gpurun merges gpurun_out/ back; elsewhere the file simply stays beside the repo's other scratch
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_detail_n%d.json" % n)
        with open(path, "w") as f:
            json.dump(out, f)
        out["detail"] = os.path.relpath(path, ROOT)
    except OSError:
        pass
    # (what libraries left in C stdio buffers -- librccl's announcement -- leaves NOW, on stderr, so that the line is also the LAST thing a reader of both streams sees)
    flush_library_text()
    print(json.dumps(final_line(out)), file=_RESULT_STREAM or sys.stdout, flush=True)


# The driver keeps rank 0's stdout, and what it wants there is ONE line.  Libraries write there too: librccl announces itself ("Librccl path : ...") through C
# stdio, which on a pipe is flushed when the process EXITS -- behind the result line, from every rank of an N > 1 run (found by tests/test_gpu_bench_flow.py; a 1-GPU
# run never loads RCCL).  So every rank hands descriptor 1 to stderr before anything is loaded, and rank 0 writes its one line to the saved descriptor.
_RESULT_STREAM = None


def claim_stdout():
    global _RESULT_STREAM
    if _RESULT_STREAM is None:
        try:
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            _RESULT_STREAM = os.fdopen(saved, "w")
        except OSError:   # (no descriptor 1 / 2 to work with: the line goes where print() sends it)
            _RESULT_STREAM = None


def flush_library_text():
    """What libraries left in C stdio buffers goes out now (to stderr, once claim_stdout() has run)."""
    try:
        sys.stdout.flush()
        sys.stderr.flush()
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


# ------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--settle-seconds", type=float, default=0.5,
                    help="untimed passes before the warm-up steps until the GPU clock has left its idle state")
    ap.add_argument("--workload", default="fir1024", choices=["fir1024", "updn43", "iir8", "fir127", "iir8tp", "iir8cas", "iirlp8", "iir8c64", "fir1024c128"] + sorted(RATE_WORKLOADS))
    ap.add_argument("--log2n", type=int, default=26, help="weak scaling: samples per GPU = 2^log2n")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--total-log2n", type=int, default=30, help="strong scaling: 2^total samples shared by all GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--other-steps", type=int, default=100)
    ap.add_argument("--no-config5", action="store_true", help="N > 1: skip the 2^30-sample strong-scaling leg behind the weak-scaling line")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="target CPU time of each baseline sample")
    ap.add_argument("--launch-timeout", type=float, default=1500.0)
    ap.add_argument("--board-seconds", type=float, default=1.5, help="0: skip the power / clock reading and the device-copy reference")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    claim_stdout()

    from sk_dsp_comm_amd import _ffi, sharding

    rank, world, local = sharding.env_rank_world()
    args.gpus = world
    tr = sharding.RcclTransport(rank, world, local)  # binds this process to GPU LOCAL_RANK
    flush_library_text()   # (every rank: librccl's announcement leaves now, not when the process exits -- behind rank 0's line for a reader of all streams as one)
    info = _ffi.device_info()

    if args.scaling == "strong":
        total = 1 << args.total_log2n
        if total % world:
            sys.exit("--scaling strong: %d GPUs do not divide 2^%d samples" % (world, args.total_log2n))
        n = total // world
    else:
        n = 1 << args.log2n
    K, W = args.steps, args.warmup

    w = make_workload(args.workload, n, rank, world, tr, _ffi, sharding)
    elapsed, ev_ms, mine = timed_steps(w, K, W, args.settle_seconds, tr, _ffi)

    board = copy_ref = None
    if world == 1 and args.board_seconds > 0:   # (before anything else touches the outputs: the parity check reads them first)
        try:
            board = board_state_of(w, args.board_seconds, _ffi)
        except Exception as e:
            board = {"skipped": "%s: %s" % (type(e).__name__, e)}
        w.step()
        _ffi.sync()

    per_rank = None
    if world > 1:
        tab = tr.allgather_state(np.array([mine[0] * 1e3 / K, mine[1] / K]))
        per_rank = {"step_ms": [float(v) for v in tab[:, 0]], "kernel_ms": [float(v) for v in tab[:, 1]]}
    n_comm = tr.comm_count()

    # quick parity spot check of what was just computed (oracle = checker only)
    check = w.check() if (rank == 0 and w.check is not None) else None
    # N > 1: EVERY rank checks the outputs that consumed its neighbour's halo / state and one interior window; the line
    # carries the maximum over ranks and the run fails (exit code 3) above the tolerance
    parity = None
    if world > 1 and w.shard is not None:
        kind, coeffs, seed = w.shard

        def get_x(g0, count):
            t = _ffi.DeviceArray(count, w.dtype).fill_noise(seed, first_index=g0)
            try:
                return t.to_host()
            finally:
                t.free()
        herr, ierr = shard_parity(kind, coeffs, n, rank, get_x, lambda i0, c: w.yd.to_host(i0, c))
        parity = reduce_parity(tr, herr, ierr, _ffi.get_option("shard_two_launches"), _ffi.get_option("shard_halo_state"))

    out = None
    if rank == 0:
        log2n = n.bit_length() - 1
        out = {
            "metric": w.metric,
            "value": float(w.units) * world * K / elapsed / 1e6,
            "unit": "MSamples/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed * 1e3 / K,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": w.arith,
            "data": "synthetic",
            "config": {"workload": w.wl, "samples_per_gpu": n, "total_samples": n * world,
                       "clock_settle_s": args.settle_seconds,
                       "sharding": "single GPU" if world == 1 else
                                   ("contiguous sample blocks, RCCL state hand-off (2 x sections doubles per hop)"
                                    if args.workload == "iir8" else
                                    "contiguous sample blocks, %d-sample RCCL halo beside the interior tiles" % (len(w.taps) - 1)
                                    if args.workload == "fir1024" else "independent replicas"),
                       "n_ranks_rccl": n_comm,
                       "device": info["name"], "compute_units": info["compute_units"]},
            "roofline": roofline_of(w, ev_ms, K, log2n),
            "kernel_source_sha256": source_hashes(args.workload),
        }
        if info["compute_units"] < 200:   # an XCD partition of the card (CPX mode: 8 logical devices of 32 CUs on ONE MI355X)
            out["config"]["logical_device"] = ("compute partition of one MI355X (%d CUs): the ranks share one HBM stack and one power "
                                               "budget -- a FUNCTIONAL run of the RCCL path, not a scaling measurement" % info["compute_units"])
        c = compute_of(w, ev_ms, K)
        if c is not None:
            out["compute"] = c
        dp = dp_pipe_of(w, ev_ms, K, info["compute_units"])
        if dp is not None:
            out["fp64_pipe"] = dp
        if board is not None:
            out["board"] = board
        if per_rank is not None:
            out["per_rank"] = per_rank
        if check is not None:
            out["parity_spot_check_max_err"] = check
        if parity is not None:
            out["parity_halo_max_err"], out["parity_interior_max_err"], out["halo_fallback_used"], out["parity_ok"], out["halo_state"] = parity
            out["halo_form"] = HALO_FORMS.get(max(out["halo_state"]) if w.shard[0] == "fir" else -1, "state hand-off (no halo)")
            out["parity_what"] = ("every rank: first 2048 outputs of its shard (they consume the %s from rank r-1) and one interior "
                                  "window vs the CPU oracle on inputs regenerated by global index; max over ranks; tolerance %g"
                                  % ("Ntaps-1 halo" if w.shard[0] == "fir" else "handed-over filter state", PARITY_TOL))

    if rank == 0 and world == 1 and args.board_seconds > 0:   # (after the parity check: the copy overwrites the outputs)
        try:
            copy_ref = device_copy_of(w, _ffi)
            out["device_copy"] = copy_ref
            out["roofline"]["frac_of_device_copy"] = out["roofline"]["achieved"] / copy_ref["GBps"]
            if copy_ref.get("own_copy_GBps"):
                out["roofline"]["frac_of_fastest_copy"] = out["roofline"]["achieved"] / max(copy_ref["GBps"], copy_ref["own_copy_GBps"])
        except Exception as e:
            out["device_copy"] = {"skipped": "%s: %s" % (type(e).__name__, e)}

    # ------------------------------------------- CPU baselines (rank 0 of a 1-GPU run only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_port(args, w)
        out["cpu_baseline_scipy"] = cpu_baseline_scipy(args, w)
    elif rank == 0:
        out["cpu_baseline"] = None

    # ------------------------------------------- the other BASELINE.json configs, same process
    if world == 1 and args.workload == "fir1024" and not args.no_other_configs and args.scaling == "weak":
        free_workload(w)
        others = {}
        for name in ("updn43", "iir8", "fir127", "iir8tp", "iir8cas", "iirlp8", "iir8c64", "fir1024c128") + OTHER_RATE_WORKLOADS:
            try:
                o = make_workload(name, 1 << 26, 0, 1, tr, _ffi, sharding)
                Ko = args.other_steps if name != "fir1024c128" else max(args.other_steps // 5, 5)
                el, ev, _ = timed_steps(o, Ko, max(10, Ko // 5), 0.1, tr, _ffi)
                r = roofline_of(o, ev, Ko, 26)
                others[name] = {"workload": o.wl, "value": float(o.units) * Ko / el / 1e6, "unit": "MSamples/s (input)",
                                "steps": Ko, "ms": el * 1e3 / Ko, "kernel_ms": r["kernel_ms"], "achieved_GBps": r["achieved"],
                                "frac": r["frac"], "traffic": r["traffic"],
                                "algorithmic_bytes_per_launch": o.alg_bytes, "traffic_source": r["traffic_source"], "kernel": o.kern}
                if getattr(o, "rate", None):   # a SURVEY 8(a) .up / .dn row: sizes at both rates
                    others[name]["rate"] = o.rate
                if o.check is not None:            # the head of the result against the oracle
                    others[name]["parity_spot_check_max_err"] = o.check()
                c = compute_of(o, ev, Ko)
                if c is not None:
                    others[name]["compute"] = c
                dp = dp_pipe_of(o, ev, Ko, info["compute_units"])
                if dp is not None:
                    others[name]["fp64_pipe"] = dp
                if args.board_seconds > 0:
                    b = board_state_of(o, min(args.board_seconds, 1.0), _ffi)
                    others[name]["board"] = {k: b.get(k) for k in ("power_w", "sclk_mhz")} if "power_w" in b else b
                free_workload(o)
            except Exception as e:  # a broken side config must not take the headline line with it
                others[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["other_configs"] = others
        rows = {"fir1024": compact_row(out["roofline"]["kernel_ms"], out["roofline"]["frac"], out["roofline"]["traffic"],
                                       out["roofline"]["algorithmic_bytes_per_launch"], out.get("parity_spot_check_max_err"), out.get("board"))}
        for name, o in others.items():
            rows[name] = ({"error": o["error"][:60]} if "error" in o else
                          compact_row(o["kernel_ms"], o["frac"], o["traffic"], o["algorithmic_bytes_per_launch"], o.get("parity_spot_check_max_err"), o.get("board")))
        out["rows_what"] = ROWS_WHAT
        out["rows"] = rows

    # ------------------------------------------- N > 1: BASELINE config 5 (2^30 samples in total, strong scaling) behind the weak-scaling line
    c5_ok = True
    if world > 1 and args.workload == "fir1024" and args.scaling == "weak" and not args.no_config5:
        free_workload(w)
        try:
            rec, c5_ok = config5_leg(args, rank, world, tr, _ffi, sharding)
        except Exception as e:   # (a failure here must not take the weak-scaling line with it -- but the run exits 3)
            rec, c5_ok = {"error": "%s: %s" % (type(e).__name__, e)}, False
        if rank == 0:
            out["config5"] = rec

    if rank == 0:
        emit(out)
    tr.close()
    if (parity is not None and not parity[3]) or not c5_ok:
        sys.exit(3)   # a wrong halo / state hand-off must not look like a result


# the SURVEY 8(a) .up / .dn rows timed behind the headline, in the order of that table
OTHER_RATE_WORKLOADS = ("upsample4", "downsample3", "firup12", "firdn12", "firup4", "firdn4", "rcup12", "rcdn12", "iirup2", "iirdn3")

# kernel sources whose change makes a committed PMC measurement stale (workload -> files under scikit-dsp-comm_amd/csrc)
TRAFFIC_SOURCES = {
    "fir1024": ["fir_ols.hip", "ols_core.hpp", "careful.hpp"], "fir127": ["fir_bx.hip", "careful.hpp"], "updn43": ["fir_bx.hip", "careful.hpp"],
    "fir1024c128": ["fir_ols64.hip", "careful.hpp"], "iir8": ["iir_par.hip"], "iirlp8": ["iir_par.hip"], "iir8c64": ["iir_par.hip"],
    "iir8cas": ["iir_fused.hip", "iir_common.hpp"], "iir8tp": ["iir_scan.hip", "iir_common.hpp"],
    "upsample4": ["resample.hip"], "downsample3": ["resample.hip"],
    "firup12": ["fir_bx.hip", "careful.hpp"], "firup4": ["fir_ols.hip", "ols_core.hpp", "careful.hpp"],
    "firdn12": ["fir_bx.hip", "careful.hpp"], "firdn4": ["fir_ols.hip", "ols_core.hpp", "careful.hpp"],
    "rcup12": ["iir_par.hip"], "rcdn12": ["iir_par.hip"], "iirup2": ["iir_par.hip"], "iirdn3": ["iir_par.hip"],
}


def _code_only(text):
    """A kernel source without its comments and white space: what the compiler sees.  The profile stamps hash THIS, so that a note added to a kernel
    does not make the measurements of the unchanged code look stale (round 6: every comment edit cost a re-collection)."""
    import re
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


def source_hashes(workload):
    import hashlib
    out = {}
    for f in TRAFFIC_SOURCES.get(workload, []):
        try:
            txt = open(os.path.join(ROOT, "scikit-dsp-comm_amd", "csrc", f), "r", errors="replace").read()
            out[f] = hashlib.sha256(_code_only(txt).encode()).hexdigest()[:16]
        except OSError:
            out[f] = None
    return out


def measured_traffic(workload, log2n):
    """(HBM bytes per step, provenance) from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE,
    WRITE_SIZE; separate --pmc runs, read side doubled per the gfx950 correction of
    MI355X_MICROARCH.md; tools/collect_profiles.sh + tools/reduce_pmc.py).  PMC counters cannot be
    read inside an un-profiled run, so this is the committed measurement of the same workload at
    the same size (the newest profiles/rNN that has one).  The profile records the hashes of the kernel sources it was
    collected with: when the sources have changed since, the number is withheld (null) instead of going stale silently."""
    if log2n != 26:
        return None, None
    prof = os.path.join(ROOT, "profiles")
    best = None
    for d in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        f = os.path.join(prof, d, "pmc_%s.json" % workload)
        if os.path.exists(f):
            best = f
    if best is None:
        return None, None
    try:
        j = json.load(open(best))
        src = {"profile": os.path.relpath(best, ROOT), "collected_at_commit": j.get("collected_at_commit"),
               "kernel_source_sha256": j.get("source_sha256")}
        if j.get("source_sha256") != source_hashes(workload):
            src["stale"] = "the kernel sources changed after this profile was collected: traffic withheld"
            return None, src
        return float(j["derived"]["hbm_total_bytes_per_step"]), src
    except Exception:
        return None, None


def _sized_sample(run, first, n_max, seconds):
    """Time run(m) on m = first samples, then once more on the m that should take `seconds`."""
    t0 = time.perf_counter()
    run(first)
    dt = max(time.perf_counter() - t0, 1e-6)
    m = int(min(n_max, max(first, first * seconds / dt)))
    if m >= 0.75 * n_max:
        m = n_max        # close enough to the whole workload: run all of it, so that the figure is not a partial-sample one
    t0 = time.perf_counter()
    run(m)
    return m, time.perf_counter() - t0


def cpu_baseline_port(args, w):
    """The oracle (C port of the reference's arithmetic: float64 accumulation of a complex64 / float32 input,
    i.e. what scipy.signal.lfilter / sosfilt do for the reference) timed on this box's host cores on a bounded
    sample of the same workload."""
    from oracle import oracle as orc
    cores_avail = os.cpu_count()
    if w.name.startswith("iir"):
        x = w.xd.to_host(0, min(w.n, 1 << 26))
        fn = orc.sos_filter if np.iscomplexobj(x) else orc.sos_filter_f32in_timed
        m, dt = _sized_sample(lambda k: fn(w.sos, x[:k]), 1 << 18, x.size, args.cpu_seconds)
        what = "sequential DF2T in float64 (sosfilt restated in C)"
    elif w.name == "updn43":
        x = w.xd.to_host(0, 1 << 20)
        m, dt = _sized_sample(lambda k: orc.downsample(orc.fir_up(w.taps, x[:k], 4), 3), 1 << 14, x.size, args.cpu_seconds)
        what = "upsample -> 512-tap FIR at the 4x rate -> downsample (the reference's three passes, restated in C)"
    else:
        x = w.xd.to_host(0, min(w.n, 1 << 24))
        fn = orc.fir_filter if x.dtype.itemsize > 8 or x.dtype == np.float64 else orc.fir_filter_f32in_timed
        m, dt = _sized_sample(lambda k: fn(w.taps, x[:k]), 1 << 16, x.size, args.cpu_seconds)
        what = "direct-form float64 accumulation (lfilter FIR branch restated in C, 1 thread like the reference)"
    return {"value": m / dt / 1e6, "unit": "MSamples/s", "cores": 1, "kind": "port",
            "sample": "first %d samples of the workload: %s" % (m, what), "cores_available": cores_avail}


def cpu_baseline_scipy(args, w):
    """The reference's own CPU path: literally the scipy.signal / NumPy calls of multirate_helper.py:104-127,
    169-192 and sigsys.py:3050-3053, 3078-3083, on a bounded sample, on this box's host cores."""
    try:
        from scipy import signal
    except Exception as e:
        return {"value": None, "kind": "scipy", "skipped": "scipy not importable on this box: %s" % e}
    if w.name.startswith("iir"):
        x = w.xd.to_host(0, min(w.n, 1 << 24))
        run = lambda k: signal.sosfilt(w.sos, x[:k])                                              # noqa: E731
        first, what = 1 << 16, "scipy.signal.sosfilt(sos, x) (multirate_helper.py:173)"
    elif w.name == "updn43":
        x = w.xd.to_host(0, 1 << 18)

        def run(k):
            xs = x[:k]
            up = np.hstack((xs.reshape(k, 1), np.zeros((k, 3)))).flatten()                        # sigsys.py:3050-3053
            y = signal.lfilter(w.taps, [1], 4 * up)                                               # multirate_helper.py:116-117
            return y[0::3]                                                                        # sigsys.py:3078-3083
        first, what = 1 << 12, "upsample (hstack/flatten) -> scipy.signal.lfilter(b,[1],4*x_up) -> strided view"
    else:
        x = w.xd.to_host(0, min(w.n, 1 << 26))   # the whole 2^26 workload when ~10 s of one host core suffice
        run = lambda k: signal.lfilter(w.taps, [1], x[:k])                                        # noqa: E731
        first, what = 1 << 14, "scipy.signal.lfilter(b, [1], x) (multirate_helper.py:108)"
    m, dt = _sized_sample(run, first, x.size, max(args.cpu_seconds, 10.0) if w.name == "fir1024" else args.cpu_seconds)
    threads = None
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        pass
    return {"value": m / dt / 1e6, "unit": "MSamples/s", "cores": 1, "kind": "scipy",
            "sample": "first %d samples of the workload through %s; the call is effectively single-threaded "
                      "(per-output dot / serial recursion)" % (m, what),
            "blas_threads_configured": threads, "cores_available": os.cpu_count()}


if __name__ == "__main__":
    main()
