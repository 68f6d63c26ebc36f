#!/bin/bash
# half-slot bf16x3 split in the M <= 8 GEMV kernels: parity of the linear kernels, then A/B of the frame against HEAD's build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear or teacher or free_run or fused" 2>&1 | tail -5 | tee gpurun_out/r5/c1_parity.txt
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 8 --frames 300 2>&1 | tee gpurun_out/r5/c1_ab_b8.txt
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 1 --frames 300 --rounds 1 2>&1 | tee gpurun_out/r5/c1_ab_b1.txt
