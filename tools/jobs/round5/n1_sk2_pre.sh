#!/bin/bash
# k_gemv_sk2: bias / residual of the k = 0 half kept apart until the epilogue (their sum at the top made the compiler wait for both before the first x / weight request)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5; O=gpurun_out/r5/n1_sk2_pre.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear or teacher or free_run" 2>&1 | grep -E "passed|failed" | tee -a $O
for B in 8 16 64; do
echo "B = $B, 300 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch $B --frames 300 --reps 3 --rounds 2 2>&1 | tee -a $O
done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tee -a $O
