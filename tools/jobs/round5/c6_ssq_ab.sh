#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
Q3TTS_LIB=$PWD/build/libq3tts_ssq.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear or teacher or free_run or fused" 2>&1 | tail -3
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so build/libq3tts_ssq.so --batch 8 --frames 300 2>&1 | tee gpurun_out/r5/c6_ab_b8.txt
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so build/libq3tts_ssq.so --batch 1 --frames 300 --rounds 1 2>&1 | tee gpurun_out/r5/c6_ab_b1.txt
