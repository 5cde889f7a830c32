#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc CSVs that tools/collect_profiles.sh wrote into
profiles/<round>/pmc_ols_tile_kernel.json (per-launch means + the derived HBM bytes that
bench.py reports as roofline.traffic).

    python tools/reduce_pmc.py gpurun_out/profiles_r01 profiles/r01

FETCH_SIZE / WRITE_SIZE are in KiB; the read side is doubled per the gfx950 correction of
MI355X_MICROARCH.md (HBM section); the passes are separate rocprofv3 runs.
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_fir1024_*.csv"))):
    if f.endswith("_trace.csv"):
        continue
    acc = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(f)):
        if "ols_tile_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in acc.items():
        out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    if dur:
        out["kernel_us_in_pass_" + os.path.basename(f)[len("pmc_fir1024_"):-4]] = sum(dur) / len(dur)
rd = out["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
wr = out["WRITE_SIZE"]["mean_per_launch"] * 1024
alg = 16 * 2 ** 26
out["derived"] = {
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_total_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg,
    "note": "FETCH_SIZE/WRITE_SIZE are in KiB; read side doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM); "
            "separate --pmc passes; kernel = skdsp::ols_tile_kernel<false,false> on 2^26 c64 samples, 1024 taps",
}
json.dump(out, open(os.path.join(dst, "pmc_ols_tile_kernel.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
