"""Per-dispatch counters of a RESIDENT launch profile (scripts/gpu_profile_config.sh ... --launch resident): one dispatch
walks K steps, so rocprofv3's per-kernel averages mix launches of different lengths; this keeps them apart.
  python scripts/resident_counters.py gpurun_out/prof_<tag> <kernel substring> <out.json>"""
import collections, csv, glob, json, os, sys

root, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]


def find(name):
    hits = glob.glob(os.path.join(root, "**", name), recursive=True)
    return hits[0] if hits else None


dispatches = collections.OrderedDict()  # order of appearance -> {counter: value, "ns": duration without counters}
trace = [r for r in csv.DictReader(open(find("trace_kernel_trace.csv"))) if kernel in r["Kernel_Name"]]
for i, r in enumerate(trace):
    dispatches[i] = {"kernel": r["Kernel_Name"].split("(")[0], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                     "vgprs": int(r["VGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"] or 0)}
for tag in ("fetch", "write", "sq", "inst"):
    path = find(f"pmc_{tag}_counter_collection.csv")
    if not path:
        continue
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for i, (_, c) in enumerate(per.items()):
        if i in dispatches:
            dispatches[i].update(c)
json.dump({"source": root, "kernel_filter": kernel, "dispatches": list(dispatches.values())}, open(out, "w"), indent=1)
for i, d in dispatches.items():
    print(i, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items()})
