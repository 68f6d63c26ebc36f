#!/bin/bash
# Builds libq3tts.so (gfx950) in-tree: qwen3_tts_rs_amd/libq3tts.so
# Incremental: an object is rebuilt when its source, one of the headers it includes, or ITS COMMAND LINE changed (the
# command is kept beside the object as <name>.cmd, so a flag change never reuses a stale object).
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
DEV_FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $PRELOAD -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value $Q3_BUILD_EXTRA"
HOST_FLAGS="-O2 -std=c++17 -fPIC -Wall"
pids=()
objs=()
# compile <source> <object> <flags> <dependency headers...>
compile() {
  local src="$1" obj="$2" flags="$3"; shift 3
  local cmd="$HIPCC $flags -c $src -o $obj" stale=0
  objs+=("$obj")
  [ -f "$obj" ] && [ -f "$obj.cmd" ] && [ "$(cat "$obj.cmd")" = "$cmd" ] || stale=1
  for dep in "$src" "$@"; do [ "$dep" -nt "$obj" ] && stale=1; done
  if [ $stale = 1 ]; then
    rm -f "$obj.cmd"
    ( $cmd && echo "$cmd" > "$obj.cmd" ) &
    pids+=($!)
  fi
}
API="$HERE/../../include/q3tts.h"
for f in q3_kernels_lm q3_kernels_gemv q3_kernels_wide q3_kernels_codec q3_kernels_prefill q3_speaker q3_mimi; do
  compile "$HERE/$f.hip" "$BUILD/$f.o" "$DEV_FLAGS" "$HERE/q3_kernels.h" "$HERE/q3_internal.h" "$HERE/q3_capture_lock.h" "$API"
done
# the engine (host side of the hot path): five units behind q3_engine.h
for f in q3_model q3_codec_run q3_session q3_batcher q3_testapi; do
  compile "$HERE/$f.hip" "$BUILD/$f.o" "$DEV_FLAGS" "$HERE/q3_engine.h" "$HERE/q3_capture_lock.h" "$HERE/q3_kernels.h" "$HERE/q3_internal.h" "$HERE/q3_aql.h" "$API"
done
compile "$HERE/q3_io.cpp" "$BUILD/q3_io.o" "$HOST_FLAGS" "$HERE/q3_internal.h" "$API"
compile "$HERE/q3_dp.cpp" "$BUILD/q3_dp.o" "$HOST_FLAGS" "$HERE/q3_internal.h" "$API"
compile "$HERE/q3_aql.cpp" "$BUILD/q3_aql.o" "$HOST_FLAGS" "$HERE/q3_aql.h" "$HERE/q3_capture_lock.h"
fail=0
for p in "${pids[@]}"; do wait $p || fail=1; done
[ $fail = 0 ] || { echo "build failed" >&2; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o "$OUT" "${objs[@]}" -ldl
echo "built $OUT"
