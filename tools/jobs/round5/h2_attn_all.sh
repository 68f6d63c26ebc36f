#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
V=attnall; L=$PWD/build/libq3tts_$V.so
Q3TTS_LIB=$L timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py tests/test_paged_kv.py -x -q -m gpu -k "not 4k and not long640 and not wide64" 2>&1 | tail -3
O=gpurun_out/r5/h2_attn_all_ab.txt
for B in 8 1; do
echo "B = $B, 300 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so $L --batch $B --frames 300 2>&1 | tee -a $O
done
echo "B = 8, 640 frames; splits" | tee -a $O
timeout 1200 python tools/dev/env_ab.py "" "Q3TTS_LIB=$L" "Q3TTS_LIB=$L Q3_ATTN_SPLITS=4" "Q3TTS_LIB=$L Q3_ATTN_SPLITS=16" --frames 640 --rounds 1 2>&1 | tee -a $O
