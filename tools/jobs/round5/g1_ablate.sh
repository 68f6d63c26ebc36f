#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
for v in "" abl1 abl2 abl4; do
  if [ -n "$v" ]; then export Q3TTS_LIB=$PWD/build/libq3tts_$v.so; else unset Q3TTS_LIB; fi
  echo "== ${v:-default}"; Q3_BENCH_M=8 timeout 300 python tools/bench_kernels.py 2>&1 | cut -c1-75
done 2>&1 | tee gpurun_out/r5/g1_ablate.txt
