#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/build/libq3tts_cpord.so; O=gpurun_out/r5/h5_cp_order_ab.txt
Q3TTS_LIB=$L timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -x -q -m gpu -k "teacher or free_run or fused or 0_6b or b8 or b16 or wide or b64" 2>&1 | tail -2 | tee -a $O
for B in 8 1 16; do
echo "B = $B, 300 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so $L --batch $B --frames 300 2>&1 | tee -a $O
done
