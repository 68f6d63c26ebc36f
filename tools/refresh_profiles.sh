#!/bin/bash
# Round-end evidence refresh (run ON the GPU box): GPU test suite, bench at B = 8 / 16, rocprofv3 kernel stats of the
# bench command, vocoder kernel table. Outputs under gpurun_out/refresh/ — copy what is to be judged into profiles/.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/refresh"; mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python bench.py --gpus 1 > "$OUT/bench_b8.json" 2> "$OUT/bench_b8.err"; tail -c 2500 "$OUT/bench_b8.json"
timeout 600 python bench.py --gpus 1 --batch 16 --no-cpu-baseline > "$OUT/bench_b16.json" 2> "$OUT/bench_b16.err"; tail -c 600 "$OUT/bench_b16.json"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o b -- python "$ROOT/bench.py" --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/rocprof_bench.log" 2>&1
f=$(find /tmp/rp -name "*kernel_stats.csv" | head -1); cp "$f" "$OUT/rocprof_kernel_stats_bench_b8.csv"; head -12 "$OUT/rocprof_kernel_stats_bench_b8.csv"
rm -rf /tmp/rp
cd "$ROOT" && bash tools/prof_vocoder.sh 640 > /dev/null 2>&1; cp "$ROOT/gpurun_out/vocprof/vocoder_T640.txt" "$OUT/vocoder_T640.txt"; head -3 "$OUT/vocoder_T640.txt"
if [ "$1" != "notests" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -3 > "$OUT/pytest_gpu.txt"; cat "$OUT/pytest_gpu.txt"
fi
