#!/bin/bash
# Matrix-core utilisation of the long-prompt prefill kernels (run ON the GPU box): one PMC pass (kernel-trace only).
# Output gpurun_out/pmc/prefill_mfma_<model>_<positions>.txt
MODEL=${1:-1.7b}; N=${2:-4096}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pq -o p -- python "$ROOT/tools/prof_prefill.py" $MODEL $N > "$OUT/prefill_mfma.log" 2>&1
python - "$MODEL" "$N" <<'PY' > "$OUT/prefill_mfma_${MODEL}_${N}.txt"
import csv, glob, sys, collections
cc = glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("/tmp/pq/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(cc)):
    d = dur.get(r["Dispatch_Id"])
    if not d or r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES": continue
    name = d[1].replace("void q3::", "").replace("q3::", "").split("(")[0]
    if not any(k in name for k in ("lm_gemm", "attn_prefill", "split_rows", "attn_merge", "row_den")): continue
    a = acc[name]; a[0] += 1; a[1] += d[0]; a[2] += float(r["Counter_Value"])
print(f"# prefill of {sys.argv[2]} instruct positions ({sys.argv[1]}), 3 prefills; util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); profiled durations")
print(f"{'kernel':40s} {'calls':>6s} {'avg us':>9s} {'ms/prefill':>11s} {'util %':>7s}")
for name, (n, ns, busy) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:40s} {n:6d} {ns / n / 1e3:9.1f} {ns / 3e6:11.2f} {busy / (ns * 2.4 * 1024) * 100:7.1f}")
PY
cat "$OUT/prefill_mfma_${MODEL}_${N}.txt"
rm -rf /tmp/pq
