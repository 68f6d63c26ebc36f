#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
V=coal4
Q3TTS_LIB=$PWD/build/libq3tts_$V.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -x -q -m gpu -k "linear or teacher or free_run or fused or 0_6b or streaming or prefill_stages" 2>&1 | tail -3
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so build/libq3tts_$V.so --batch 1 --frames 300 2>&1 | tee gpurun_out/r5/g4_coal4_ab.txt
