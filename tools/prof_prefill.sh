#!/bin/bash
# Per-kernel breakdown of a long-prompt prefill (run ON the GPU box): prof_prefill.sh <model> <positions>
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prefillprof"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python "$ROOT/tools/prof_prefill.py" ${1:-1.7b} ${2:-4096} > "$OUT/run.log" 2>&1
grep "prefill positions" "$OUT/run.log"
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if any(k in r["Name"] for k in ("lm_gemm", "attn_prefill", "row_den", "attn_fused", "gemv", "rope", "gather"))]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:6]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e6/3:8.2f} ms/prefill avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
rm -rf /tmp/pp
