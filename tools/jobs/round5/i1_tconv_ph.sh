#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
L=$PWD/build/libq3tts_ph.so; O=gpurun_out/r5/i1_tconv_ph.txt
for PH in 0 1 2 3; do
echo "== Q3_TCONV_PH=$PH" | tee -a $O
Q3TTS_LIB=$L Q3_TCONV_PH=$PH timeout 600 python -m pytest tests/test_bench_config_parity.py -x -q -m gpu -k "full_size_vocoder" 2>&1 | tail -2 | tee -a $O
Q3TTS_LIB=$L Q3_TCONV_PH=$PH bash tools/prof_vocoder.sh 640 >/dev/null 2>&1
grep -E "one decode|<2, " gpurun_out/vocprof/vocoder_T640.txt | head -6 | tee -a $O
Q3TTS_LIB=$L Q3_TCONV_PH=$PH python tools/prof_decode.py 640 3 2>&1 | tail -1 | tee -a $O
done
