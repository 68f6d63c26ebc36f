#!/bin/bash
# Counter comparison for one GEMV shape (run ON the GPU box): pmc_probe.sh <name N K epi rms M> — product kernel vs the
# -DQ3_ABLATE=2 build (no x / norm loads), a few counters per pass.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmcprobe"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in prod ablate2; do
  if [ $lib = ablate2 ]; then export Q3TTS_LIB="$ROOT/build/libq3tts_ablate2.so"; else unset Q3TTS_LIB; fi
  for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
    tag=$(echo $ctrs | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OUT/${lib}_$tag" -o p -- python "$ROOT/tools/pmc_gemv.py" "$@" > "$OUT/${lib}_$tag.log" 2>&1
  done
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUTDIR", "") or "."
res = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath("x")), "x"))): pass
PY
cd "$OUT" && python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("*/p_counter_collection.csv")):
    lib = f.split("_")[0]
    for r in csv.DictReader(open(f)):
        if "gemv" in r["Kernel_Name"]:
            res[r["Counter_Name"]][lib].append(float(r["Counter_Value"]))
print(f"{'counter':34s} {'prod':>14s} {'no-x ablate':>14s} {'ratio':>7s}")
for c, d in sorted(res.items()):
    a = sum(d["prod"]) / max(len(d["prod"]), 1) if d.get("prod") else float("nan")
    b = sum(d["ablate2"]) / max(len(d["ablate2"]), 1) if d.get("ablate2") else float("nan")
    print(f"{c:34s} {a:14.1f} {b:14.1f} {a / b if b else float('nan'):7.2f}")
PY
find "$OUT" -name "*kernel_trace.csv" -delete
