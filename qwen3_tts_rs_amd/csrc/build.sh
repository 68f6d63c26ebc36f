#!/bin/bash
# Builds libq3tts.so (gfx950) in-tree: qwen3_tts_rs_amd/libq3tts.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
# A/B aids: Q3_BUILD_OUT / Q3_BUILD_DIR = another library / object directory, Q3_BUILD_EXTRA = extra hipcc flags,
# Q3_NO_PRELOAD=1 = without kernel-argument preload
OUT="${Q3_BUILD_OUT:-$HERE/../libq3tts.so}"
BUILD="${Q3_BUILD_DIR:-$HERE/../../build}"
mkdir -p "$BUILD"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
PRELOAD="-mllvm -amdgpu-kernarg-preload-count=14"
[ -n "$Q3_NO_PRELOAD" ] && PRELOAD=""
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $PRELOAD -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value $Q3_BUILD_EXTRA"
pids=()
for f in q3_kernels_lm q3_kernels_gemv q3_kernels_wide q3_kernels_codec q3_kernels_prefill q3_engine q3_speaker q3_mimi; do
  if [ ! -f "$BUILD/$f.o" ] || [ "$HERE/$f.hip" -nt "$BUILD/$f.o" ] || [ "$HERE/q3_kernels.h" -nt "$BUILD/$f.o" ] || [ "$HERE/../../include/q3tts.h" -nt "$BUILD/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$BUILD/$f.o" &
    pids+=($!)
  fi
done
if [ ! -f "$BUILD/q3_io.o" ] || [ "$HERE/q3_io.cpp" -nt "$BUILD/q3_io.o" ] || [ "$HERE/q3_internal.h" -nt "$BUILD/q3_io.o" ] || [ "$HERE/../../include/q3tts.h" -nt "$BUILD/q3_io.o" ]; then
  $HIPCC -O2 -std=c++17 -fPIC -Wall -c "$HERE/q3_io.cpp" -o "$BUILD/q3_io.o" &
  pids+=($!)
fi
if [ ! -f "$BUILD/q3_dp.o" ] || [ "$HERE/q3_dp.cpp" -nt "$BUILD/q3_dp.o" ] || [ "$HERE/q3_internal.h" -nt "$BUILD/q3_dp.o" ] || [ "$HERE/../../include/q3tts.h" -nt "$BUILD/q3_dp.o" ]; then
  $HIPCC -O2 -std=c++17 -fPIC -Wall -c "$HERE/q3_dp.cpp" -o "$BUILD/q3_dp.o" &
  pids+=($!)
fi
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o "$OUT" "$BUILD/q3_kernels_lm.o" "$BUILD/q3_kernels_gemv.o" "$BUILD/q3_kernels_wide.o" "$BUILD/q3_kernels_codec.o" "$BUILD/q3_kernels_prefill.o" "$BUILD/q3_engine.o" "$BUILD/q3_speaker.o" "$BUILD/q3_mimi.o" "$BUILD/q3_io.o" "$BUILD/q3_dp.o" -ldl
echo "built $OUT"
