#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
O=gpurun_out/r5/h7_b16_bisect.txt
B=$PWD/build/libq3tts_base.so; N=$PWD/qwen3_tts_rs_amd/libq3tts.so
for K in "" "Q3_NO_CP_ATTN=1" "Q3_NO_KSPLIT=1" "Q3_CP_NO_FOLD=1"; do
timeout 1200 python tools/dev/env_ab.py "Q3TTS_LIB=$B $K" "Q3TTS_LIB=$N $K" --batch 16 --frames 100 --reps 1 --rounds 2 2>&1 | sed "s#$PWD/##g" | tee -a $O
done
for BB in 12 9; do
timeout 600 python tools/dev/env_ab.py "Q3TTS_LIB=$B" "Q3TTS_LIB=$N" --batch $BB --frames 100 --reps 1 --rounds 1 2>&1 | sed "s#$PWD/##g" | tee -a $O
done
