#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sampler or frame_embed or free_run or eos or teacher or rows_with" 2>&1 | tail -3 | tee gpurun_out/r5/c3_parity.txt
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 8 --frames 300 2>&1 | tee gpurun_out/r5/c3_ab_b8.txt
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 1 --frames 300 --rounds 1 2>&1 | tee gpurun_out/r5/c3_ab_b1.txt
