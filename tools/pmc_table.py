"""Per-kernel means of every counter in one or more rocprofv3 --pmc CSV runs (development aid):
pmc_table.py <substring of kernel names to keep> <dir> [<dir> ...]"""
import csv, glob, sys, collections
pat = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for d in sys.argv[2:]:
    for cc in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(cc)):
            name = r["Kernel_Name"].replace("void q3::", "").replace("q3::", "").split("(")[0]
            if pat not in name: continue
            key = (name, int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
            a = agg[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
            did = (cc, r["Dispatch_Id"])
            if did not in seen and "End_Timestamp" in r:
                seen.add(did); dd = dur[key]; dd[0] += 1; dd[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for key in sorted(agg, key=lambda k: -dur[k][1]):
    n, t = dur[key]
    print(f"{key[0][:60]} grid {key[1]}  calls {n}  avg {t / max(n, 1) / 1e3:.1f} us")
    for c, (m, v) in sorted(agg[key].items()):
        print(f"    {c:28s} {v / m:16.0f}")
