"""Reduce the FETCH_SIZE / WRITE_SIZE passes of `tools/time_fir_up.py 4x256 12x256` (complex64) to profiles/rNN/pmc_fir_up.txt:
    python tools/reduce_pmc_up.py gpurun_out/profiles_r03 profiles/r03
expects <in>/pmc_fir_up_FETCH_SIZE.csv and <in>/pmc_fir_up_WRITE_SIZE.csv (tools/collect_profiles.sh writes them)."""
import collections, csv, os, shutil, statistics, sys

src, dst = sys.argv[1], sys.argv[2]
labels = ["L = 4  (input 128 MiB)   strided stores", "L = 4  (input 128 MiB)   rows + weave", "L = 4  default (= strided)",
          "L = 12 (input 42.7 MiB)  strided stores", "L = 12 (input 42.7 MiB)  rows + weave", "L = 12 default (= rows + weave)"]
out = ["rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- DTYPES=complex64 python tools/time_fir_up.py 4x256 12x256",
       "complex64, 2^26 outputs (512 MiB); per launch, MiB; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM).",
       "The tool runs, per shape, 13 launches each of: polyphase kernels, walk with strided stores, walk as rows + weave, default dispatch.",
       "raw counters: pmc_fir_up_FETCH_SIZE.csv, pmc_fir_up_WRITE_SIZE.csv"]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = os.path.join(src, "pmc_fir_up_%s.csv" % c)
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = int(r["Dispatch_Id"])
        acc.setdefault(d, [r["Kernel_Name"], 0.0])
        acc[d][1] += float(r["Counter_Value"] or 0)
    seq = [(n, v / 1024 * (2 if c == "FETCH_SIZE" else 1)) for d, (n, v) in acc.items() if "ols_tile" in n or "interleave" in n]
    forms, i = [], 0
    while i < len(seq):   # a walk launch directly followed by the weaving copy = the rows form
        n, v = seq[i]
        if "ols_tile" in n and i + 1 < len(seq) and "interleave" in seq[i + 1][0]:
            forms.append((v, seq[i + 1][1])); i += 2
        elif "ols_tile" in n:
            forms.append((v, None)); i += 1
        else:
            i += 1
    out += ["", c + " (MiB per launch: median, min .. max of 13 launches)"]
    for k in range(0, len(forms), 13):
        items = forms[k:k + 13]
        w = [x[0] for x in items]
        line = "  %-42s walk %7.1f  (%7.1f .. %7.1f)" % (labels[k // 13] if k // 13 < len(labels) else "", statistics.median(w), min(w), max(w))
        if items[0][1] is not None:
            line += "   weave %6.1f" % statistics.median([x[1] for x in items])
        out.append(line)
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
open(os.path.join(dst, "pmc_fir_up.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
