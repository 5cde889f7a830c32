"""The per-workload table of profiles/<round>/README.md from the committed artefacts: kernel-trace average, un-profiled bench line, PMC traffic,
board reading, joules per launch.  python tools/profiles_table.py profiles/r04"""
import csv, glob, json, os, sys
d = sys.argv[1]
rows = []
order = ["fir1024", "updn43", "iir8", "fir127", "fir1024c128", "iir8c64", "iirlp8", "iir8cas", "iir8tp",
         "upsample4", "downsample3", "firup12", "firdn12", "firup4", "firdn4", "rcup12", "rcdn12", "iirup2", "iirdn3"]
print("| workload | kernel under the trace (avg of ALL launches / min us, launches) | steady state under the trace: mean / median us of the timed steps (HIP events of the same run) -> % of 8 TB/s | un-profiled ms | % of 8 TB/s | HBM traffic / algorithmic (MB read + written) | board W / shader MHz | mJ per launch |")
print("|---|---|---|---|---|---|---|---|")
for w in order:
    try:
        b = json.loads(open(os.path.join(d, "bench_%s.json" % w)).read().strip().splitlines()[-1])
    except Exception:
        continue
    ks = list(csv.DictReader(open(os.path.join(d, "kernel_stats_%s.csv" % w))))
    ks = [r for r in ks if "noise_" not in r["Name"] and "copyBuffer" not in r["Name"] and "fillBuffer" not in r["Name"]]
    k = ks[0]
    name = k["Name"].replace("(anonymous namespace)::", "").replace("void skdsp::", "").split("(")[0]
    name = name.replace("HIP_vector_type<float, 2u>", "float2").replace("HIP_vector_type<double, 2u>", "double2")
    pm = json.load(open(os.path.join(d, "pmc_%s.json" % w)))["derived"]
    board = b.get("board") or {}
    ms = b["ms_per_step"]
    try:
        st = json.load(open(os.path.join(d, "kernel_steady_%s.json" % w)))
        steady = "%.1f / %.1f (%.1f) -> %.1f" % (st["steady_mean_us"], st["steady_median_us"], 1e3 * st["same_run_bench_line"]["kernel_ms_hip_events"],
                                                 100 * st["frac_of_8TBps_from_steady_mean"])
    except Exception:
        steady = "-"
    print("| %s | `%s` %.1f / %.1f (%s) | %s | %.4f | %.1f | %.3f (%.1f + %.1f) | %s / %s | %.0f |" % (
        w, name, float(k["AverageNs"]) / 1e3, float(k["MinNs"]) / 1e3, k["Calls"], steady, ms, 100 * b["roofline"]["frac"],
        pm["traffic_over_algorithmic"], pm["hbm_read_bytes_per_step"] / 1e6, pm["hbm_write_bytes_per_step"] / 1e6,
        "%.0f" % board.get("power_w", 0), "%.0f" % board.get("sclk_mhz", 0), (board.get("power_w") or 0) * ms))
