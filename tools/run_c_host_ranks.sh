#!/bin/bash
# N ranks of the torch-free C host (examples/c_host.c), one per GPU, over the library's own RCCL path (q3_dp_*): the
# data-parallel start-up without Python in any process — rank 0 publishes the RCCL id through a file, one weight broadcast,
# every rank synthesises its utterance, timings are all-gathered, rank 0 prints the job's frames/s.
#   tools/run_c_host_ranks.sh N [out_dir] [frames]
# Needs N visible GPUs (RCCL refuses two ranks per device: gpurun_out/native_rccl_world2_same_gpu.json).
set -e
N="${1:?usage: run_c_host_ranks.sh N [out_dir] [frames]}"; OUT="${2:-gpurun_out/c_host_ranks}"; FRAMES="${3:-32}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$OUT" "$ROOT/build"
gcc -O1 -I"$ROOT/include" "$ROOT/examples/c_host.c" -o "$ROOT/build/c_host" -L"$ROOT/qwen3_tts_rs_amd" -lq3tts -Wl,-rpath,"$ROOT/qwen3_tts_rs_amd"
RDV="$(mktemp -d)"; trap 'rm -rf "$RDV"' EXIT          # a fresh directory per launch: no stale id can be read
export HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for ((r = 0; r < N; r++)); do
  Q3_RANK=$r Q3_WORLD=$N Q3_ID_FILE="$RDV/rccl_id" "$ROOT/build/c_host" "$OUT" "$FRAMES" > "$OUT/rank$r.log" 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
cat "$OUT/rank0.log"
[ $rc = 0 ] || { echo "a rank failed:"; tail -n 5 "$OUT"/rank*.log; exit 1; }
