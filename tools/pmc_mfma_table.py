"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace CSV run (development aid):
pmc_mfma_table.py <dir> <runs> <title>. util = busy cycles / (duration x 2.4 GHz x 1024 SIMDs), profiled durations."""
import csv, glob, sys, collections
d, runs, title = sys.argv[1], float(sys.argv[2]), sys.argv[3]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES": continue
    name = r["Kernel_Name"].replace("void q3::", "").replace("q3::", "").split("(")[0]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else 0
    a = agg[name]; a[0] += 1; a[1] += dur; a[2] += float(r["Counter_Value"])
print(f"# {title}; util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); profiled durations")
print(f"{'kernel':44s} {'calls':>6s} {'avg us':>9s} {'ms/run':>9s} {'util %':>7s}")
for name, (n, dur, busy) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    util = 100.0 * busy / (dur * 2.4 * 1024) if dur else 0.0
    print(f"{name[:44]:44s} {n:6d} {dur / n / 1e3:9.1f} {dur / 1e6 / runs:9.2f} {util:7.1f}")
