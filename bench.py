#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X streaming-filter path.

    python bench.py --gpus N --steps K --warmup W           (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W    (N>1, one rank per GPU)

A "step" = one pass of multirate_FIR.filter (1024-tap lowpass, complex64) over each
rank's contiguous sample block of 2^26 samples, inputs already resident in HBM:
halo exchange of the Ntaps-1 = 1023 preceding samples over RCCL (N>1), then the
overlap-save kernel.  value = total samples all ranks filtered / max-over-ranks time.

The product path uses no PyTorch; with N>1 the ranks rendezvous through a file
(sk_dsp_comm_amd.sharding.FileRendezvous) and barrier / max-reduce through RCCL.

Prints ONE JSON line on rank 0 (fields per the driver contract + "roofline" and
"cpu_baseline").  Other workloads (--workload updn43|iir8|fir127) are measurement aids
for the remaining BASELINE.json configs and print the same shape.
"""
import argparse
import json
import os
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--settle-seconds", type=float, default=0.15,
                    help="untimed passes before the warm-up steps until the GPU clock has left its idle state")
    ap.add_argument("--workload", default="fir1024", choices=["fir1024", "updn43", "iir8", "fir127"])
    ap.add_argument("--log2n", type=int, default=26, help="samples per GPU = 2^log2n")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    args = ap.parse_args()

    from sk_dsp_comm_amd import _ffi, sharding

    rank, world, local = sharding.env_rank_world()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with one process per GPU "
                     "(python -m torch.distributed.run --nproc-per-node %d ...)" % (args.gpus, args.gpus))
        args.gpus = world
    tr = sharding.RcclTransport(rank, world, local)  # binds this process to GPU LOCAL_RANK
    info = _ffi.device_info()

    n = 1 << args.log2n
    K, W = args.steps, args.warmup

    # ------------------------------------------------------------- workload
    if args.workload == "fir1024":
        b = firwin_lowpass(1024, 0.2)
        dtype, arith = np.complex64, "c64"
        fir = sharding.ShardedFIR(b, tr, dtype=dtype)
        xd = fir.new_shard_buffer(n).fill_noise(2026, first_index=rank * n)
        yd = _ffi.DeviceArray(n, dtype)
        step = lambda: fir.filter_local_dev(xd, yd, n)           # noqa: E731
        units, alg_bytes = n, 16.0 * n                           # 8 B in + 8 B out per sample
        kern = "ols_tile_kernel"
        wl = "multirate_FIR.filter: 1024-tap lowpass, complex64, 2^%d samples per GPU, FFT overlap-save" % args.log2n
        metric = "complex64 MSamples/s (FIR-1024 tap, 2^26 samples)"
    elif args.workload == "fir127":
        b = firwin_lowpass(127, 0.2)
        dtype, arith = np.float32, "f32"
        k = _ffi.FirKernel(b, _ffi.F32)
        xd = _ffi.DeviceArray(n, dtype).fill_noise(2026)
        yd = _ffi.DeviceArray(n, dtype)
        step = lambda: k.filter_dev(xd, yd)                      # noqa: E731
        units, alg_bytes = n, 8.0 * n
        kern = "fir_bx_kernel"
        wl = "multirate_FIR.filter: 127-tap lowpass, float32, 2^%d samples, direct form on the BF16 matrix pipe (3-way bf16 split = float32 precision)" % args.log2n
        metric = "float32 MSamples/s (FIR-127 tap)"
    elif args.workload == "updn43":
        b = firwin_lowpass(512, 0.225)
        dtype, arith = np.complex64, "c64"
        k = _ffi.FirKernel(b, _ffi.C64)
        xd = _ffi.DeviceArray(n, dtype).fill_noise(2026)
        n_out = (n * 4) // 3
        yd = _ffi.DeviceArray(n_out, dtype)
        step = lambda: k.updn_dev(xd, yd, 4, 3)                  # noqa: E731
        units, alg_bytes = n, 8.0 * n + 8.0 * n_out              # 18.67 B per input sample
        # float32 arithmetic carried by 6 bf16 products per multiply: against the FP32 matrix / vector peak the useful
        # flops may exceed 100 % -- that is the point of the split
        compute = ("useful f32 flops vs the FP32 matrix-pipe peak (computed as 6 bf16 MFMA products per multiply)", 157.3, 4.0 * 512 / 4 * n_out)  # 4*Ntaps/L flop per c64 output
        kern = "fir_bx_kernel"
        wl = "downsample(multirate_FIR.up(x,4),3): 512-tap prototype, complex64, 2^%d input samples, fused polyphase (Toeplitz product on the BF16 matrix pipe, 3-way bf16 split = float32 precision)" % args.log2n
        metric = "complex64 input MSamples/s (polyphase L=4/M=3, 512 taps)"
    else:
        sos = elliptic_bpf_sos()
        dtype, arith = np.float32, "f32 I/O, f64 state"
        xd = _ffi.DeviceArray(n, dtype).fill_noise(2026, first_index=rank * n)
        yd = _ffi.DeviceArray(n, dtype)
        if world == 1:
            k = _ffi.IirKernel(_ffi.F32, sos=sos)
            step = lambda: k.filter_dev(xd, yd)                  # noqa: E731
        else:  # contiguous sample blocks, exact state hand-off rank r -> r+1 (16 doubles per hop)
            iir = sharding.ShardedIIR(sos, tr, dtype=dtype)
            step = lambda: iir.filter_local_dev(xd, yd, n)       # noqa: E731
        units, alg_bytes = n, 8.0 * n
        compute = ("FP64 vector (v_fma_f64)", 78.6, 72.0 * n)     # 9 flop per biquad per sample
        kern = "iir_k1r_kernel + iir_carry_kernel + iir_chunk_kernel"
        wl = "multirate_IIR.filter: 8-biquad elliptic bandpass, float32, 2^%d samples, affine scan" % args.log2n
        metric = "float32 MSamples/s (8-biquad SOS IIR)"

    if args.workload in ("fir1024", "fir127"):
        compute = None
    # --------------------------------------------------------------- timing
    # The chip idles at ~160 MHz; the first ~50 launches after idle run on a ramping clock (0.28 ms
    # for the first 50-launch window of the headline kernel, 0.236 ms from the second window on,
    # flat for as long as the launches continue: profiles/r01/README.md).  Steady state is what a
    # streaming job sees, so the clock is settled first, untimed, whatever W is.
    # (the number of settle passes is agreed between the ranks: every pass of a sharded workload
    # contains a send/recv pair, so a per-rank time-based loop would deadlock)
    t_settle = time.perf_counter()
    for _ in range(10):
        step()
    _ffi.sync()
    per_pass = max((time.perf_counter() - t_settle) / 10, 1e-6)
    n_settle = int(tr.allreduce_max(float(min(20000, int(args.settle_seconds / per_pass) + 1)))) if args.settle_seconds > 0 else 0
    for _ in range(n_settle):
        step()
    _ffi.sync()
    for _ in range(W):
        step()
    _ffi.sync()
    tr.barrier()
    _ffi.timer_start()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    ev_ms = _ffi.timer_stop()  # HIP events on the stream the kernels run on (synchronises)
    _ffi.sync()
    tr.barrier()
    t1 = time.perf_counter()
    elapsed = tr.allreduce_max(t1 - t0)
    ev_ms = tr.allreduce_max(ev_ms)

    # quick parity spot check of what was just computed (oracle = checker only)
    check = None
    if rank == 0 and args.workload == "fir1024":
        from oracle import oracle as orc
        s0, w = 3 * 7168 - 100, 2048
        xs = xd.to_host(s0 - 1023, w + 1023)
        ref = orc.fir_filter(b, xs)[1023:]
        got = yd.to_host(s0, w)
        check = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))

    # --------------------------------------------------------- cpu baseline
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, b if args.workload != "iir8" else None, xd)

    if rank == 0:
        total_units = float(units) * world * K
        ms_per_step = elapsed * 1e3 / K
        t_kernel = ev_ms * 1e-3 / K  # average launch (+ halo) duration from HIP events
        achieved = alg_bytes / t_kernel / 1e9
        traffic = measured_traffic(args)
        out = {
            "metric": metric,
            "value": total_units / elapsed / 1e6,
            "unit": "MSamples/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": arith,
            "data": "synthetic",
            "config": {"workload": wl, "samples_per_gpu": n, "total_samples": n * world, "clock_settle_s": args.settle_seconds,
                       "sharding": "single GPU" if world == 1 else
                                   ("contiguous sample blocks, RCCL state hand-off (2 x sections doubles per hop)"
                                    if args.workload == "iir8" else
                                    "contiguous sample blocks, %d-sample RCCL halo" % (len(b) - 1)
                                    if args.workload == "fir1024" else "independent replicas"),
                       "device": info["name"], "compute_units": info["compute_units"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": kern,
                         "kernel_ms": t_kernel * 1e3, "algorithmic_bytes_per_launch": alg_bytes},
            "cpu_baseline": cpu,
        }
        if compute is not None:
            # these two workloads are bound by vector arithmetic, not by HBM: useful
            # flops of the reference formulation against the vector peak (the IIR scan executes 2.2x them)
            tf = compute[2] / t_kernel / 1e12
            out["compute"] = {"unit": "TFLOP/s", "what": compute[0], "useful_flop_per_step": compute[2], "achieved": tf,
                              "peak": compute[1], "frac": tf / compute[1]}
        if check is not None:
            out["parity_spot_check_max_err"] = check
        print(json.dumps(out))
    tr.close()


def measured_traffic(args):
    """HBM bytes per step from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE,
    WRITE_SIZE; separate --pmc runs, read side doubled per the gfx950 correction of
    MI355X_MICROARCH.md; tools/collect_profiles.sh + tools/reduce_pmc.py).  PMC counters cannot be
    read inside an un-profiled run, so this is the committed measurement of the same workload at
    the same size, or null when none applies."""
    if args.log2n != 26:
        return None
    prof = os.path.join(ROOT, "profiles")
    best = None
    for d in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        f = os.path.join(prof, d, "pmc_%s.json" % args.workload)
        if os.path.exists(f):
            best = f
    if best is None:
        return None
    try:
        return float(json.load(open(best))["derived"]["hbm_total_bytes_per_step"])
    except Exception:
        return None


def cpu_baseline(args, b, xd):
    """The oracle (C port of the reference's arithmetic: float64 accumulation of a
    complex64/float32 input, i.e. what scipy.signal.lfilter/sosfilt do for the reference)
    timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    cores_avail = os.cpu_count()
    if args.workload == "iir8":
        sos = elliptic_bpf_sos()
        m = 1 << 22
        x = xd.to_host(0, m)
        t0 = time.perf_counter(); orc.sos_filter_f32in_timed(sos, x[:1 << 18]); dt = time.perf_counter() - t0
        m = int(min(1 << 26, max(1 << 18, (1 << 18) * args.cpu_seconds / max(dt, 1e-6))))
        m = min(m, xd.n)
        x = xd.to_host(0, m)
        best = min(_timed(lambda: orc.sos_filter_f32in_timed(sos, x)) for _ in range(2))
        return {"value": m / best / 1e6, "unit": "MSamples/s", "cores": 1, "kind": "port",
                "sample": "first %d of the 2^%d float32 samples, sequential DF2T in float64 (sosfilt restated in C)" % (m, args.log2n),
                "cores_available": cores_avail}
    if args.workload == "updn43":
        m = 1 << 16
        x = xd.to_host(0, m)
        best = min(_timed(lambda: orc.downsample(orc.fir_up(b, x, 4), 3)) for _ in range(2))
        return {"value": m / best / 1e6, "unit": "MSamples/s", "cores": 1, "kind": "port",
                "sample": "first %d input samples through upsample -> 512-tap FIR at the 4x rate -> downsample" % m,
                "cores_available": cores_avail}
    x = xd.to_host(0, 1 << 16)
    t0 = time.perf_counter(); orc.fir_filter_f32in_timed(b, x); dt = time.perf_counter() - t0
    m = int(min(xd.n, max(1 << 16, (1 << 16) * args.cpu_seconds / max(dt, 1e-6))))
    x = xd.to_host(0, m)
    best = min(_timed(lambda: orc.fir_filter_f32in_timed(b, x)) for _ in range(2))
    return {"value": m / best / 1e6, "unit": "MSamples/s", "cores": 1, "kind": "port",
            "sample": "first %d of the 2^%d samples, direct-form float64 accumulation (lfilter FIR branch restated in C, 1 thread like the reference)" % (m, args.log2n),
            "cores_available": cores_avail}


def _timed(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


if __name__ == "__main__":
    main()
