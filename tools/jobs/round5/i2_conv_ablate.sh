#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
O=gpurun_out/r5/i2_conv_ablate.txt
for V in base abl1 abl2 abl3; do
L=$PWD/build/libq3tts_$V.so; [ $V = base ] && L=$PWD/qwen3_tts_rs_amd/libq3tts.so
echo "== $V" | tee -a $O
Q3TTS_LIB=$L bash tools/prof_vocoder.sh 640 >/dev/null 2>&1
grep -E "one decode|k_conv_bf16x3<[27], " gpurun_out/vocprof/vocoder_T640.txt | head -10 | tee -a $O
done
