#!/bin/bash
# round 6 / B4: the engine as five translation units — the whole -m gpu suite; +residual instances of k_gemv_mfma4 with their
# residual / bias pointers in the preloaded argument slots: A/B on single-utterance frames (1.7B and 0.6B) against the commit before
# (build/libq3tts_base.so = HEAD of the split commit); the default bench line.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 1 --frames 300 > gpurun_out/r6/b4_mfma4_preload_ab.txt 2>&1
python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --model 0.6b --batch 1 --frames 300 >> gpurun_out/r6/b4_mfma4_preload_ab.txt 2>&1
python tools/dev/lib_ab.py build/libq3tts_base.so qwen3_tts_rs_amd/libq3tts.so --batch 8 --frames 300 >> gpurun_out/r6/b4_mfma4_preload_ab.txt 2>&1
cat gpurun_out/r6/b4_mfma4_preload_ab.txt
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6/b4_suite.txt 2>&1
tail -6 gpurun_out/r6/b4_suite.txt
timeout 900 python bench.py > gpurun_out/r6/b4_bench.json 2> gpurun_out/r6/b4_bench.err
head -c 300 gpurun_out/r6/b4_bench.json; echo
