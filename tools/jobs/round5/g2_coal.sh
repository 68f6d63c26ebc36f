#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
V=${1:-coal}
Q3TTS_LIB=$PWD/build/libq3tts_$V.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear or teacher or free_run or fused" 2>&1 | tail -3
for v in "" $V; do
  if [ -n "$v" ]; then export Q3TTS_LIB=$PWD/build/libq3tts_$v.so; else unset Q3TTS_LIB; fi
  echo "== ${v:-default}"; Q3_BENCH_M=8 timeout 300 python tools/bench_kernels.py 2>&1 | cut -c1-75
done 2>&1 | tee gpurun_out/r5/g2_$V.txt
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so build/libq3tts_$V.so --batch 8 --frames 300 2>&1 | tee gpurun_out/r5/g2_${V}_ab_b8.txt
