"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count / mean duration, and the idle gaps
between consecutive kernels of the steady-state frame loop. Usage: prof_analyze.py <kernel_trace.csv> [frames]"""
import csv, sys, collections, json
path = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)))
rows.sort()
# steady state = the graph-replayed frame loop: from the third sampler launch to the last one (one k_sample per frame);
# traces without a sampler fall back to the longest run of kernels whose gaps are all < 100 us
samp = [i for i, r in enumerate(rows) if "k_sample" in r[2]]
if len(samp) > 8:
    seg = rows[samp[2] + 1:samp[-1] + 1]
else:
    best = (0, 0); i0 = 0
    for i in range(1, len(rows) + 1):
        if i == len(rows) or rows[i][0] - rows[i - 1][1] > 100_000:
            if i - i0 > best[1] - best[0]:
                best = (i0, i)
            i0 = i
    seg = rows[best[0]:best[1]]
span = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, *_ in seg)
gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
print(f"steady segment: {len(seg)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({100*busy/span:.1f}%), "
      f"mean gap {sum(gaps)/len(gaps)/1e3:.2f} us, median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us")
nsamp = sum(1 for r in seg if "k_sample" in r[2])
if nsamp:
    frames = nsamp          # one sampler launch per frame: the steady segment's true frame count
if frames:
    print(f"per frame ({frames} frames in the segment): {len(seg)/frames:.1f} kernels, {span/frames/1e6:.3f} ms span, {busy/frames/1e6:.3f} ms busy")
agg = collections.defaultdict(lambda: [0, 0, 0])
for i, (s, e, name, g, w) in enumerate(seg):
    short = name.replace("(anonymous namespace)::", "").replace("void q3::", "").replace("q3::", "").replace("void ", "").split("(")[0]
    a = agg[(short, g // max(w, 1), w)]
    a[0] += 1; a[1] += e - s
    if i + 1 < len(seg):
        a[2] += seg[i + 1][0] - e      # gap AFTER this kernel
print(f"{'kernel':44s} {'WGs':>6s} {'thr':>5s} {'calls':>8s} {'/frame':>7s} {'avg us':>8s} {'gap us':>7s} {'ms/frame':>9s} {'%busy':>6s}")
for (short, wgs, w), (n, t, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{short:44s} {wgs:6d} {w:5d} {n:8d} {n/max(frames,1):7.1f} {t/n/1e3:8.2f} {gp/n/1e3:7.2f} {(t+gp)/max(frames,1)/1e6:9.4f} {100*t/busy:6.2f}")
# Q3_PROF_SEQ=1: the launch order of ONE frame from the middle of the steady segment (duration and the gap after each launch)
import os
if os.environ.get("Q3_PROF_SEQ") and nsamp > 4:
    si = [i for i, r in enumerate(seg) if "k_sample" in r[2]]
    a0, a1 = si[len(si) // 2] , si[len(si) // 2 + 1]
    print(f"\none frame, launch order (k_sample .. the launch before the next k_sample): {a1 - a0} launches")
    for i in range(a0, a1):
        s, e, name, g, w = seg[i]
        short = name.replace("(anonymous namespace)::", "").replace("void q3::", "").replace("q3::", "").replace("void ", "").split("(")[0]
        print(f"{i - a0:4d} {short:46s} {g // max(w, 1):5d} WGs {(e - s)/1e3:7.2f} us  gap {(seg[i + 1][0] - e)/1e3:6.2f}")
