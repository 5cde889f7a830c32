"""Per-launch sums of the counters in a rocprofv3 counter_collection.csv, by kernel: python tools/pmc_summary.py <csv> [substr]"""
import csv, sys, collections
f = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sub not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    print(k[:90], "launches", len(disp[k]))
    for c, val in sorted(v.items()): print("   %-34s %.5g" % (c, val / len(disp[k])))
