#!/bin/bash
# A/B of the frame loop between library builds (run ON the GPU box): ab_frame.sh <lib.so> [B] [frames]
for lib in "" "$1"; do
  if [ -n "$lib" ]; then export Q3TTS_LIB="$lib"; fi
  for rep in 1 2; do python tools/prof_run.py 1.7b ${2:-8} ${3:-300} | tail -1 | sed "s|^|${lib:-default}: |"; done
done
