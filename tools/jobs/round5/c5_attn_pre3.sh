#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
Q3TTS_LIB=$PWD/build/libq3tts_pre3.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_paged_kv.py -x -q -m gpu -k "teacher or free_run or paged or prefill_stages or bf16" 2>&1 | tail -3
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so build/libq3tts_pre3.so --batch 8 --frames 640 --reps 2 2>&1 | tee gpurun_out/r5/c5_ab_b8.txt
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so build/libq3tts_pre3.so --batch 1 --frames 300 --rounds 1 2>&1 | tee gpurun_out/r5/c5_ab_b1.txt
