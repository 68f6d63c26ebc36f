#!/bin/bash
# Builds libq3tts.so of a git ref beside the working tree's (for tools/dev/lib_ab.py):  build_base.sh [ref] -> build/libq3tts_base.so
set -e
REF="${1:-HEAD}"; ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
rm -rf "$ROOT/build/base"; mkdir -p "$ROOT/build/base"
git -C "$ROOT" archive "$REF" qwen3_tts_rs_amd/csrc include | tar -x -C "$ROOT/build/base"
Q3_BUILD_OUT="$ROOT/build/libq3tts_base.so" Q3_BUILD_DIR="$ROOT/build/base/obj" bash "$ROOT/build/base/qwen3_tts_rs_amd/csrc/build.sh"
