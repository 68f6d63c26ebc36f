#!/bin/bash
# B = 1: code-predictor attention inside the o-projection (k_gemv4_attn): identity against the two-launch form, B = 1 parity tests, A/B of the frame
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5; O=gpurun_out/r5/k1_attn_in_o.txt; : > $O
timeout 900 python -m pytest tests/test_kernel_variants.py -x -q -m gpu -k single_row 2>&1 | tail -15 | tee -a $O
for B in 1 2; do
echo "B = $B, 300 frames" | tee -a $O
timeout 900 python tools/dev/env_ab.py "" "Q3_CP_NO_ATTN_FOLD=1" --batch $B --frames 300 --reps 3 --rounds 2 2>&1 | sed "s#$PWD/##g" | tee -a $O
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -x -q -m gpu -k "not 4k and not long640 and not wide64" 2>&1 | tail -5 | tee -a $O
