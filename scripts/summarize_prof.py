"""Condense rocprofv3 CSV outputs (kernel stats + PMC passes) into a short text summary for profiles/."""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]


def rows(pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


st = rows("trace_kernel_stats.csv")
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for r in st:
    print(f"{r.get('Name','?')[:70]:70s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")
for tag, ctr in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_sq", None), ("pmc_inst", None)):
    rs = rows(f"{tag}_counter_collection.csv")
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for r in rs:
        k = r.get("Kernel_Name", "?")
        agg[k][r.get("Counter_Name")] += float(r.get("Counter_Value", 0))
    disp = defaultdict(set)
    for r in rs:
        disp[r.get("Kernel_Name", "?")].add(r.get("Dispatch_Id"))
    print(f"== {tag} (per-dispatch averages) ==")
    for k, cs in agg.items():
        n = max(len(disp[k]), 1)
        if "nam_" not in k:
            continue
        print(f"{k[:60]:60s} dispatches={n} " + " ".join(f"{c}={v / n:.4g}" for c, v in sorted(cs.items())))
