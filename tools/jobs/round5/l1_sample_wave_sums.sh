#!/bin/bash
# k_sample: the sequential sums (top-p, final softmax + CDF) walked over a wave's lanes instead of one thread's LDS loop: sampler parity, codes, A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5; O=gpurun_out/r5/l1_sample_wave_sums.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sampler or eos or free_run" 2>&1 | tail -4 | tee -a $O
for B in 8 1; do
echo "B = $B, 300 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch $B --frames 300 --reps 3 --rounds 2 2>&1 | tee -a $O
done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $O
