#!/bin/bash
# round 6 / D1: the wide-session family (k_wide_gemm, k_wide_epilogue, k_wide2_split, k_wide2_swiglu) on the write-through transport,
# its packets without fences: A/B against the commit before (build/libq3tts_base.so) at 64 and 32 utterances; wide-session tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 64 --frames 300 > gpurun_out/r6/d1_wide_ab.txt 2>&1
python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 32 --frames 300 >> gpurun_out/r6/d1_wide_ab.txt 2>&1
python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 20 --frames 300 --rounds 1 >> gpurun_out/r6/d1_wide_ab.txt 2>&1
cat gpurun_out/r6/d1_wide_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "b32_b64 or wide or variants or b8_b16 or config3 or frame_submission" > gpurun_out/r6/d1_tests.txt 2>&1; tail -4 gpurun_out/r6/d1_tests.txt
