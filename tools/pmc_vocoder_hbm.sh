#!/bin/bash
# HBM traffic of the vocoder kernels (run ON the GPU box): FETCH_SIZE and WRITE_SIZE in separate passes (kernel-trace
# only), gfx950 x2 correction on FETCH_SIZE (MI355X_MICROARCH.md §HBM). Output gpurun_out/pmc/vocoder_hbm_T<T>.txt
T=${1:-640}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ph_$ctr
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/ph_$ctr -o p -- python "$ROOT/tools/prof_decode.py" $T 1 > "$OUT/vocoder_hbm_$ctr.log" 2>&1
done
python - "$T" <<'PY' > "$OUT/vocoder_hbm_T$T.txt"
import csv, glob, sys, collections
T = sys.argv[1]
acc = collections.defaultdict(lambda: {"n": 0, "ns": 0.0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    cc = glob.glob(f"/tmp/ph_{ctr}/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(f"/tmp/ph_{ctr}/**/*kernel_trace.csv", recursive=True)[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1)) for r in csv.DictReader(open(kt))}
    for r in csv.DictReader(open(cc)):
        d = dur.get(r["Dispatch_Id"])
        if not d or r["Counter_Name"] != ctr or "conv" not in d[1] and "attn_c" not in d[1] and "norm_c" not in d[1]: continue
        name = d[1].replace("void q3::", "").replace("q3::", "").split("(")[0]
        a = acc[(name, d[2])]
        a[ctr] += float(r["Counter_Value"]) * 1024 * (2 if ctr == "FETCH_SIZE" else 1)
        if ctr == "FETCH_SIZE": a["n"] += 1; a["ns"] += d[0]
print(f"# vocoder decode T = {T}; per-launch means; FETCH_SIZE x2 (gfx950), KiB units; rate = (fetch + write) / profiled duration")
print(f"{'kernel':44s} {'WGx':>6s} {'calls':>6s} {'avg us':>9s} {'fetch MB':>10s} {'write MB':>10s} {'GB/s':>8s}")
for (name, wgs), a in sorted(acc.items(), key=lambda kv: -kv[1]["ns"]):
    n = max(a["n"], 1)
    f, w, us = a["FETCH_SIZE"] / n, a["WRITE_SIZE"] / n, a["ns"] / n / 1e3
    print(f"{name:44s} {wgs:6d} {a['n']:6d} {us:9.1f} {f / 1e6:10.1f} {w / 1e6:10.1f} {(f + w) / us / 1e3:8.0f}")
PY
head -28 "$OUT/vocoder_hbm_T$T.txt"
rm -rf /tmp/ph_FETCH_SIZE /tmp/ph_WRITE_SIZE
