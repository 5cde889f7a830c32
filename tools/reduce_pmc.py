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
    for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_INSTS_MFMA"):
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
    out["source_sha256"] = bench.source_hashes(w)     # (uncommitted edits at collection time show up as a hash no commit has)
    out["collected_at_commit"] = COMMIT
    json.dump(out, open(os.path.join(dst, "pmc_%s.json" % w), "w"), indent=1)
    print(w, "traffic/algorithmic = %.3f  (%.1f MB read + %.1f MB written per step)" % ((rd + wr) / alg, rd / 1e6, wr / 1e6))

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
