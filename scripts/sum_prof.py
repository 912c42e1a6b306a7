"""rocprofv3 counter CSVs -> per-STEP totals of one kernel: the counters of every dispatch whose name contains KERNEL are
summed and divided by STEPS (a persistent session spreads one run's steps over a few launches of uneven length, so
per-dispatch averages mean nothing). Usage: python scripts/sum_prof.py <dir> <kernel substring> <steps>"""
import csv, glob, os, sys
from collections import defaultdict

d, kernel, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
tot = defaultdict(float)
ndisp = defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if kernel in r.get("Kernel_Name", ""):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                ndisp[r["Counter_Name"]].add((os.path.basename(f), r["Dispatch_Id"]))
dur = 0.0
n = 0
for f in glob.glob(os.path.join(d, "**", "trace_kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if kernel in r.get("Kernel_Name", ""):
                dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                n += 1
print(f"== {kernel}: totals / {steps:.0f} steps ==")
print(f"dispatches={n} kernel_ns_per_step={dur / steps:.1f}")
print(" ".join(f"{k}={v / steps:.5g}" for k, v in sorted(tot.items())))
print("dispatch counts per counter:", {k: len(v) for k, v in ndisp.items()})
