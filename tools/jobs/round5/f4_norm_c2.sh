#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py tests/test_speech_encoder.py -q -x -m gpu -k "vocoder or decoder or seamless or residual or streaming or speech or mimi or segment or decode" 2>&1 | tail -3
bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
head -1 gpurun_out/vocprof/vocoder_T640.txt; grep -E "k_norm_c" gpurun_out/vocprof/vocoder_T640.txt | head -4
