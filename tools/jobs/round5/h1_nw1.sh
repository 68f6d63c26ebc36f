#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
V=nw1
Q3TTS_LIB=$PWD/build/libq3tts_$V.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -x -q -m gpu -k "linear or teacher or free_run or fused or 0_6b or streaming or prefill_stages" 2>&1 | tail -3
for B in 8 4; do
timeout 900 python tools/dev/lib_ab.py qwen3_tts_rs_amd/libq3tts.so build/libq3tts_$V.so --batch $B --frames 300 2>&1 | tee -a gpurun_out/r5/h1_nw1_ab.txt
done
