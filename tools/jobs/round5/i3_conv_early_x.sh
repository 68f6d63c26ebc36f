#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/build/libq3tts_earlyx.so; O=gpurun_out/r5/i3_conv_early_x.txt
Q3TTS_LIB=$L timeout 900 python -m pytest tests/test_bench_config_parity.py tests/test_gpu_parity.py -x -q -m gpu -k "vocoder or codec or streaming" 2>&1 | grep -E "passed|failed" | tee -a $O
for V in base earlyx base earlyx; do
LL=$L; [ $V = base ] && LL=$PWD/qwen3_tts_rs_amd/libq3tts.so
echo "== $V" | tee -a $O
Q3TTS_LIB=$LL bash tools/prof_vocoder.sh 640 >/dev/null 2>&1
grep -E "one decode|k_conv_bf16x3<[27], 1, 4" gpurun_out/vocprof/vocoder_T640.txt | head -7 | tee -a $O
done
