#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
O=gpurun_out/r5/h6_rn_helpers.txt
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $O
for B in 8 1 16; do
echo "B = $B, 300 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch $B --frames 300 2>&1 | tee -a $O
done
