#!/bin/bash
# k_gemv_mfma: bias / residual requests behind the first group's x / weight requests (their addresses need the un-preloaded argument struct): A/B, parity, suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5; O=gpurun_out/r5/o1_pre_behind_first_group.txt; : > $O
for B in 8 1; do
echo "B = $B, 300 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch $B --frames 300 --reps 3 --rounds 2 2>&1 | tee -a $O
done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tee -a $O
