#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
V=coal16
Q3TTS_LIB=$PWD/build/libq3tts_$V.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -x -q -m gpu -k "linear or teacher or free_run or fused or b8_b16 or b32_b64 or wide_session" 2>&1 | tail -3
for B in 16 64; do
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so build/libq3tts_$V.so --batch $B --frames 200 --reps 2 2>&1 | sed "s/^/B=$B /"
done | tee gpurun_out/r5/g3_coal16_ab.txt
