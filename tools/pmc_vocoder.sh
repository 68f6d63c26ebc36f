#!/bin/bash
# Matrix-core utilisation of the vocoder kernels (run ON the GPU box): SQ_VALU_MFMA_BUSY_CYCLES per kernel against
# duration x 1024 SIMDs x clock. One PMC pass (kernel-trace only, as gpurun requires), output gpurun_out/pmc/vocoder_mfma_T<T>.txt
T=${1:-640}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pv -o p -- python "$ROOT/tools/prof_decode.py" $T 2 > "$OUT/vocoder_mfma.log" 2>&1
python - "$T" <<'PY' > "$OUT/vocoder_mfma_T$T.txt"
import csv, glob, sys, collections
T = sys.argv[1]
cc = glob.glob("/tmp/pv/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("/tmp/pv/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1))
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in csv.DictReader(open(cc)):
    d = dur.get(r["Dispatch_Id"])
    if not d: continue
    name = d[1].replace("void q3::", "").replace("q3::", "").split("(")[0]
    a = acc[(name, d[2])]
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": a[0] += 1; a[1] += d[0]; a[2] += float(r["Counter_Value"])
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE": a[3] += float(r["Counter_Value"])
print(f"# vocoder decode, T = {T} frames, 2 decodes; util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); profiled durations")
print(f"{'kernel':44s} {'WGs':>6s} {'calls':>6s} {'avg us':>9s} {'mfma busy Mcyc':>15s} {'gui Mcyc':>10s} {'util %':>7s}")
tot_b = tot_d = 0.0
for (name, wgs), (n, ns, busy, gui) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if n == 0: continue
    util = busy / (ns * 2.4 * 1024) * 100
    tot_b += busy; tot_d += ns
    print(f"{name:44s} {wgs:6d} {n:6d} {ns / n / 1e3:9.1f} {busy / 1e6:15.2f} {gui / 1e6:10.2f} {util:7.1f}")
print(f"all kernels: MFMA busy {tot_b / (tot_d * 2.4 * 1024) * 100:.1f} % of SIMD cycles over {tot_d / 1e6:.2f} ms")
PY
head -30 "$OUT/vocoder_mfma_T$T.txt"
rm -rf /tmp/pv
