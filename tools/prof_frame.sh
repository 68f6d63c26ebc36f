#!/bin/bash
# Per-kernel breakdown of the steady-state frame loop (run ON the GPU box): prof_frame.sh <model> <B> <frames> [prompt]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/frameprof"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
Q3_PROMPT=${4:-700} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/fp -o t -- python "$ROOT/tools/prof_run.py" $1 $2 $3 > "$OUT/run_$1_b$2.log" 2>&1
f=$(find /tmp/fp -name "*kernel_trace.csv" | head -1)
python "$ROOT/tools/prof_analyze.py" "$f" $3 > "$OUT/frame_$1_b$2.txt" 2>&1
tail -2 "$OUT/run_$1_b$2.log"; cat "$OUT/frame_$1_b$2.txt"
rm -rf /tmp/fp
