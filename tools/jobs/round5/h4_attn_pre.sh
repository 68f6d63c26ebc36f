#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py tests/test_paged_kv.py tests/test_ragged_batch.py -x -q -m gpu 2>&1 | tail -3
O=gpurun_out/r5/h4_attn_pre_ab.txt
echo "B = 8, 640 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 8 --frames 640 2>&1 | tee -a $O
for B in 16 64; do
echo "B = $B, 300 frames" | tee -a $O
timeout 900 python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch $B --frames 300 --rounds 1 2>&1 | tee -a $O
done
