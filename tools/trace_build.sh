#!/bin/bash
# Development build with -DQ3_TRACE (q3_kernels.h): qwen3_tts_rs_amd/libq3tts_trace.so, objects under build/trace/.
# Select it with Q3TTS_LIB=qwen3_tts_rs_amd/libq3tts_trace.so (tools/trace_frame.py does).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; SRC="$ROOT/qwen3_tts_rs_amd/csrc"; B="$ROOT/build/trace"; mkdir -p "$B"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=14 -DQ3_TRACE -Wno-unused-function -Wno-unused-variable -Wno-unused-value"
pids=()
for f in q3_kernels_lm q3_kernels_gemv q3_kernels_wide q3_kernels_codec q3_kernels_prefill q3_model q3_codec_run q3_session q3_batcher q3_testapi q3_speaker q3_mimi; do
  $HIPCC $FLAGS -c "$SRC/$f.hip" -o "$B/$f.o" & pids+=($!)
done
$HIPCC -O2 -std=c++17 -fPIC -c "$SRC/q3_io.cpp" -o "$B/q3_io.o" & pids+=($!)
$HIPCC -O2 -std=c++17 -fPIC -c "$SRC/q3_dp.cpp" -o "$B/q3_dp.o" & pids+=($!)
$HIPCC -O2 -std=c++17 -fPIC -c "$SRC/q3_aql.cpp" -o "$B/q3_aql.o" & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
OBJS=""; for f in q3_kernels_lm q3_kernels_gemv q3_kernels_wide q3_kernels_codec q3_kernels_prefill q3_model q3_codec_run q3_session q3_batcher q3_testapi q3_speaker q3_mimi q3_io q3_dp q3_aql; do OBJS="$OBJS $B/$f.o"; done      # (not "$B"/*.o: a stale object of a removed unit would be linked in)
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o "$ROOT/qwen3_tts_rs_amd/libq3tts_trace.so" $OBJS -ldl
echo "built $ROOT/qwen3_tts_rs_amd/libq3tts_trace.so"
