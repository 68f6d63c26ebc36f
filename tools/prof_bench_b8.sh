#!/bin/bash
# The committed kernel table of the HEADLINE workload (run ON the GPU box): rocprofv3 --kernel-trace of
#   python bench.py --headline-only --steps 1 --warmup 0        (1.7B, B = 8, 512-token prompts, 640 frames; nothing else runs)
# summarised over the steady segment of the frame loop (third to last sampler launch) by tools/prof_analyze.py:
# kernels per frame, mean duration, mean gap to the next kernel, ms per frame. Output: gpurun_out/<name> (default
# r4_rocprof_kernel_stats_bench_b8.txt), to be copied to profiles/. Extra arguments go to bench.py (e.g. --batch 64).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="${Q3_PROF_NAME:-r4_rocprof_kernel_stats_bench_b8.txt}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fpb
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/fpb -o t -- python "$ROOT/bench.py" --headline-only --steps 1 --warmup 0 "$@" > "$ROOT/gpurun_out/prof_bench_run.log" 2>&1
f=$(find /tmp/fpb -name "*kernel_trace.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace of: python bench.py --headline-only --steps 1 --warmup 0 $* (tools/prof_bench_b8.sh); steady segment of the frame loop only"
  echo "# bench line of the same (profiled) run: $(grep '^{' "$ROOT/gpurun_out/prof_bench_run.log" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print({k: d[k] for k in ("value","ms_per_step","stage_ms")})' 2>/dev/null)"
  python "$ROOT/tools/prof_analyze.py" "$f" 640
} > "$ROOT/gpurun_out/$NAME" 2>&1
cat "$ROOT/gpurun_out/$NAME" | head -40
rm -rf /tmp/fpb
