"""Per-kernel means of rocprofv3 PMC counters over one run (development aid): pmc_kernels.py <dir> -> table of
kernel (name, grid), calls, avg us, and every counter found (mean per launch). Durations are the profiled ones."""
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: {"n": collections.Counter(), "v": collections.Counter(), "d": 0.0, "seen": set()})
names = set()
for r in csv.DictReader(open(cc)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void q3::", "").replace("q3::", "").replace("void ", "").split("(")[0]
    key = (name, int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) // max(int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1), 1))
    a = agg[key]; c = r["Counter_Name"]; names.add(c)
    a["n"][c] += 1; a["v"][c] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in a["seen"]:
        a["seen"].add(r["Dispatch_Id"]); a["d"] += dur.get(r["Dispatch_Id"], 0)
names = sorted(names)
print(f"{'kernel':40s} {'WGs':>6s} {'calls':>6s} {'avg us':>8s} " + " ".join(f"{c[:22]:>22s}" for c in names))
for (name, wgs), a in sorted(agg.items(), key=lambda kv: -kv[1]["d"])[:24]:
    n = len(a["seen"])
    print(f"{name[:40]:40s} {wgs:6d} {n:6d} {a['d'] / max(n, 1) / 1e3:8.2f} " + " ".join(f"{(a['v'][c] / a['n'][c]) if a['n'][c] else 0:22.1f}" for c in names))
