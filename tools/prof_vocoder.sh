#!/bin/bash
# Per-kernel breakdown of one whole-utterance vocoder decode (run ON the GPU box): prof_vocoder.sh [T=640]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/vocprof"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/vp -o t -- python "$ROOT/tools/prof_decode.py" ${1:-640} 3 > "$OUT/run.log" 2>&1
f=$(find /tmp/vp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > "$OUT/vocoder_T${1:-640}.txt"
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1), int(r["Workgroup_Size_X"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# last third = the last of 1 warm + 3 timed decodes... keep kernels after the last "k_rvq_embed"
starts = [i for i, r in enumerate(rows) if "k_rvq_embed" in r[2]]
seg = rows[starts[-1]:]
span = seg[-1][1] - seg[0][0]; busy = sum(e - s for s, e, *_ in seg)
print(f"one decode: {len(seg)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms")
agg = collections.OrderedDict()
for s, e, name, g, w in seg:
    short = name.replace("(anonymous namespace)::", "").replace("void q3::", "").replace("q3::", "").replace("void ", "").split("(")[0]
    k = (short, g // max(w, 1))
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
print(f"{'kernel':40s} {'WGs':>8s} {'calls':>6s} {'total ms':>9s} {'avg us':>9s} {'%':>6s}")
for (short, wgs), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{short:40s} {wgs:8d} {n:6d} {t/1e6:9.3f} {t/n/1e3:9.1f} {100*t/busy:6.2f}")
print("\nlaunch order (launches > 40 us; Q3_VOC_MIN_NS=0 lists all):")
for i, (s, e, name, g, w) in enumerate(seg):
    if e - s > int(__import__("os").environ.get("Q3_VOC_MIN_NS", "40000")):
        short = name.replace("(anonymous namespace)::", "").replace("void q3::", "").replace("q3::", "").replace("void ", "").split("(")[0]
        print(f"{i:4d} {short:60s} {g // max(w, 1):7d} WGs {(e - s)/1e3:9.1f} us  gap {((s - seg[i-1][1]) if i else 0)/1e3:6.1f}")
PY
tail -1 "$OUT/run.log"; cat "$OUT/vocoder_T${1:-640}.txt"
rm -rf /tmp/vp
