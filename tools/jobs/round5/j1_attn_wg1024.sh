#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/build/libq3tts_wg1024.so; O=gpurun_out/r5/j1_attn_wg1024.txt
Q3TTS_LIB=$L Q3_ATTN_SPLITS=1 Q3_ATTN_WG1024=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py tests/test_paged_kv.py -x -q -m gpu -k "not 4k and not long640 and not wide64" 2>&1 | grep -E "passed|failed" | tee -a $O
for B in 8 16; do
echo "B = $B, 640 frames" | tee -a $O
timeout 1500 python tools/dev/env_ab.py "" "Q3TTS_LIB=$L" "Q3TTS_LIB=$L Q3_ATTN_SPLITS=1" "Q3TTS_LIB=$L Q3_ATTN_SPLITS=1 Q3_ATTN_WG1024=1" "Q3TTS_LIB=$L Q3_ATTN_SPLITS=1 Q3_ATTN_WG1024=2" "Q3TTS_LIB=$L Q3_ATTN_SPLITS=1 Q3_ATTN_WG1024=3" --batch $B --frames 640 --reps 2 --rounds 1 2>&1 | sed "s#$PWD/##g" | tee -a $O
done
