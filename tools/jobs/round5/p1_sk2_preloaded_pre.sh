#!/bin/bash
# k_gemv_sk2: residual / bias pointers and the residual pitch in the preloaded argument slots this family does not use (W2, norm weight, epi): A/B, suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5; O=gpurun_out/r5/p1_sk2_preloaded_pre.txt; : > $O
echo "B = 8, 300 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 8 --frames 300 --reps 3 --rounds 2 2>&1 | tee -a $O
for B in 16 64; do
echo "B = $B, 300 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch $B --frames 300 --reps 2 --rounds 1 2>&1 | tee -a $O
done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tee -a $O
