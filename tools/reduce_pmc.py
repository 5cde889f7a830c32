#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc CSVs that tools/collect_profiles.sh wrote into
profiles/<round>/pmc_<workload>.json: per-dispatch means per kernel plus the HBM bytes of ONE
bench step (what bench.py reports as roofline.traffic).

    python tools/reduce_pmc.py gpurun_out/profiles_r01 profiles/r01 [workload ...]

FETCH_SIZE / WRITE_SIZE are in KiB; the read side is doubled per the gfx950 correction of
MI355X_MICROARCH.md (HBM section); every counter group comes from its own rocprofv3 run.
A "step" is one bench.py pass: one dispatch of the FIR kernels, K1 + K3 for the IIR.
"""
import collections
import csv
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
ONLY = set(sys.argv[3:])   # (workloads re-collected after a kernel changed: only their files are reduced and stamped -- the stamp is the tree's at reduce time)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subprocess
import bench  # TRAFFIC_SOURCES / source_hashes: the profile records which kernel sources it was collected with
try:
    COMMIT = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE).stdout.decode().strip() or None
except Exception:
    COMMIT = None
WORK = {  # workload -> (substring of every kernel of a step, substring of the kernel that runs once per step, algorithmic bytes)
    "fir1024": ("ols_tile_kernel", "ols_tile_kernel", 16 * 2 ** 26),
    "fir127": ("skdsp::fir_", "skdsp::fir_", 8 * 2 ** 26),
    "updn43": ("skdsp::fir_", "skdsp::fir_", 8 * 2 ** 26 + 8 * ((2 ** 26 * 4) // 3)),
    "iir8": ("skdsp::iir_par", "skdsp::iir_par", 8 * 2 ** 26),   # config 4: the parallel-form single-pass scan (the default since round 3)
    "iir8cas": ("skdsp::iir_fused", "skdsp::iir_fused", 8 * 2 ** 26),   # ... forced through the cascade-form single pass (round 2's default)
    "fir1024c128": ("ols64_tile_kernel", "ols64_tile_kernel", 32 * 2 ** 26),
    "iir8tp": ("skdsp::iir_", "float, true", 8 * 2 ** 26),  # ... forced through K1 (matrix pipe) + carries + K3 (the WRITE = true instantiation runs once per step)
    "iir8c64": ("skdsp::iir_par", "skdsp::iir_par", 16 * 2 ** 26),   # config 4's filter on an interleaved complex64 signal
    "iirlp8": ("skdsp::iir_par", "skdsp::iir_par", 8 * 2 ** 26),   # rate_change(12)'s lowpass, parallel-form single-pass scan
}
# the .up / .dn rows of SURVEY.md 8(a) (bench.RATE_WORKLOADS, 2^26 samples at the high rate): every library kernel of a step counts (one launch
# per step, whatever engine AUTO takes), algorithmic bytes as bench.py counts them
_LIB = ("up2k_kernel", "up4k_kernel", "dn4k_kernel", "skdsp::fir_", "ols_tile_kernel", "ols_fold_kernel", "ols_rep_kernel", "interleave_kernel", "skdsp::iir_par", "upsample_kernel", "downsample_kernel", "downsample_tile_kernel")
for _w in bench.OTHER_RATE_WORKLOADS:
    _R = {"upsample4": 4, "downsample3": 3, "firup12": 12, "firdn12": 12, "firup4": 4, "firdn4": 4, "rcup12": 12, "rcdn12": 12, "iirup2": 2, "iirdn3": 3}[_w]
    _up = _w in ("upsample4", "firup12", "firup4", "rcup12", "iirup2")
    _esz = 8 if _w in ("upsample4", "downsample3", "firup12", "firdn12", "firup4", "firdn4") else 4
    _nin = 2 ** 26 // _R if _up else 2 ** 26
    _nout = _nin * _R if _up else _nin // _R
    WORK[_w] = (_LIB, _LIB, _esz * (_nin + _nout))


def _has(pat, name):
    return any(q in name for q in pat) if isinstance(pat, tuple) else pat in name


for w, (pat, marker, alg) in WORK.items():
    if ONLY and w not in ONLY:
        continue
    out = {}
    for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_INSTS_MFMA"):   # (the first counter of each pass names its file)
        f = os.path.join(src, "pmc_%s_%s.csv" % (w, tag))
        if not os.path.exists(f):
            continue
        acc = collections.defaultdict(list)
        steps = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if _has(pat, r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if _has(marker, r["Kernel_Name"]):
                    steps[r["Counter_Name"]] += 1
        for k, v in acc.items():
            out[k] = {"dispatches": len(v), "steps": steps[k], "per_step": sum(v) / max(steps[k], 1)}
    if "FETCH_SIZE" not in out or "WRITE_SIZE" not in out:
        continue
    rd = out["FETCH_SIZE"]["per_step"] * 1024 * 2
    wr = out["WRITE_SIZE"]["per_step"] * 1024
    out["derived"] = {
        "hbm_read_bytes_per_step": rd, "hbm_write_bytes_per_step": wr, "hbm_total_bytes_per_step": rd + wr,
        "algorithmic_bytes_per_step": alg, "traffic_over_algorithmic": (rd + wr) / alg,
        "note": "FETCH_SIZE/WRITE_SIZE in KiB; read side doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM); "
                "separate --pmc passes of `bench.py --workload %s`; all kernels matching %r summed per bench step" % (w, pat),
    }
    if "SQ_WAVES" in out and out["SQ_WAVES"]["per_step"] > 0:   # the issue / wait picture per wave (the *_CYCLES counters of a wave are in quad-cycles)
        wv = out["SQ_WAVES"]["per_step"]
        out["per_wave"] = {k[3:].lower(): out[k]["per_step"] / wv for k in out if k.startswith("SQ_") and k != "SQ_WAVES"}
        pw = out["per_wave"]
        if "wave_cycles" in pw and pw["wave_cycles"] > 0:
            out["per_wave"]["fraction_of_wave_life"] = {k: pw[k] / pw["wave_cycles"] for k in ("active_inst_any", "wait_inst_any", "wait_any", "active_inst_valu", "active_inst_lds") if k in pw}
    out["source_sha256"] = bench.source_hashes(w)     # (uncommitted edits at collection time show up as a hash no commit has)
    out["collected_at_commit"] = COMMIT
    json.dump(out, open(os.path.join(dst, "pmc_%s.json" % w), "w"), indent=1)
    print(w, "traffic/algorithmic = %.3f  (%.1f MB read + %.1f MB written per step)" % ((rd + wr) / alg, rd / 1e6, wr / 1e6))

# ---- the steady-state kernel duration of every workload, from the per-dispatch records of its kernel trace ------------------------------
# kernel_stats_<w>.csv averages over EVERY launch of the run -- the ~0.5 s of clock-settle launches included, which run on a ramping clock
# (round 5: avg 242.3 us, min 227.3, max 348.4 over 732 calls of the headline kernel, against 233.7 un-profiled).  bench.py times its LAST
# `steps` launches; so does this: the mean and the median of the last `steps` dispatches of the step's kernel(s), and the roofline fraction
# that follows from them.  The traced run's own bench line (HIP events around the same launches) is kept beside it.
def _steady(w, pat, marker, alg):
    f = os.path.join(src, "kernel_trace_%s.csv" % w)
    if not os.path.exists(f):
        return None
    rows = []
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if _has(pat, name) and "fill_noise" not in name:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    if not rows:
        return None
    line = None
    try:
        line = json.loads(open(os.path.join(src, "trace_bench_%s.json" % w)).read().strip().splitlines()[-1])
    except Exception:
        pass
    steps = int(line["steps"]) if line else 400
    # a step may be several launches (the two-pass IIR): group by the marker kernel, which runs once per step
    per_step, cur = [], 0.0
    for st, en, name in rows:
        cur += (en - st) * 1e-3
        if _has(marker, name):
            per_step.append(cur)
            cur = 0.0
    timed = per_step[-steps:] if len(per_step) >= steps else per_step
    timed_sorted = sorted(timed)
    mean = sum(timed) / len(timed)
    med = timed_sorted[len(timed_sorted) // 2]
    out = {"workload": w, "steps_in_trace": len(per_step), "timed_steps": len(timed),
           "all_launches_avg_us": sum(per_step) / len(per_step), "steady_mean_us": mean, "steady_median_us": med,
           "steady_min_us": timed_sorted[0], "steady_max_us": timed_sorted[-1],
           "algorithmic_bytes_per_step": alg, "frac_of_8TBps_from_steady_mean": alg / (mean * 1e-6) / 8e12,
           "what": "rocprofv3 --kernel-trace of `bench.py --workload %s`: the last %d steps (the ones bench.py times), kernel time summed per step" % (w, len(timed))}
    if line:
        out["same_run_bench_line"] = {"kernel_ms_hip_events": line["roofline"]["kernel_ms"], "frac": line["roofline"]["frac"], "ms_per_step": line["ms_per_step"]}
    return out


for w, (pat, marker, alg) in WORK.items():
    if ONLY and w not in ONLY:
        continue
    st = _steady(w, pat, marker, alg)
    if st is None:
        continue
    st["source_sha256"] = bench.source_hashes(w)
    st["collected_at_commit"] = COMMIT
    json.dump(st, open(os.path.join(dst, "kernel_steady_%s.json" % w), "w"), indent=1)
    print(w, "steady %.1f us (median %.1f, all launches %.1f) -> %.3f of 8 TB/s%s" % (
        st["steady_mean_us"], st["steady_median_us"], st["all_launches_avg_us"], st["frac_of_8TBps_from_steady_mean"],
        "; same run, HIP events: %.1f us" % (1e3 * st["same_run_bench_line"]["kernel_ms_hip_events"]) if "same_run_bench_line" in st else ""))

# the kernel traces of the same collection get the same stamp (which sources, which commit), next to the CSV
for f in sorted(os.listdir(src)):
    if f.startswith("kernel_stats_") and f.endswith(".csv"):
        w = f[len("kernel_stats_"):-4]
        if ONLY and w not in ONLY:
            continue
        if os.path.abspath(src) != os.path.abspath(dst):
            import shutil
            shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
        meta = {"workload": w, "source_sha256": bench.source_hashes(w), "collected_at_commit": COMMIT,
                "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload %s --no-cpu-baseline --no-other-configs --board-seconds 0" % w}
        json.dump(meta, open(os.path.join(dst, "kernel_stats_%s.meta.json" % w), "w"), indent=1)
