#!/bin/bash
# One source file rebuilt with extra flags and linked with the default build's other objects (for A/B builds of one kernel file):
#   build_variant.sh <name> <file-stem, e.g. q3_kernels_codec> "<extra hipcc flags>"   ->  build/libq3tts_<name>.so
set -e
NAME="$1"; STEM="$2"; EXTRA="$3"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; SRC="$ROOT/qwen3_tts_rs_amd/csrc"; B="$ROOT/build"
[ -f "$B/$STEM.o" ] || { echo "run csrc/build.sh first"; exit 1; }
mkdir -p "$B/var_$NAME"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=14 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value $EXTRA \
  -c "$SRC/$STEM.hip" -o "$B/var_$NAME/$STEM.o"
objs=()
for o in "$B"/*.o; do
  if [ "$(basename "$o")" = "$STEM.o" ]; then objs+=("$B/var_$NAME/$STEM.o"); else objs+=("$o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "$B/libq3tts_$NAME.so" "${objs[@]}" -ldl
echo "built $B/libq3tts_$NAME.so"
