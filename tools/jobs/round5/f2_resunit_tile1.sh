#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
export Q3TTS_LIB=$PWD/build/libq3tts_tile32.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -q -x -m gpu -k "residual_unit or vocoder or seamless" 2>&1 | tail -3
for v in tile32; do
  bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
  echo "== $v: $(head -1 gpurun_out/vocprof/vocoder_T640.txt) | $(grep k_resunit gpurun_out/vocprof/vocoder_T640.txt | head -1)"
done 2>&1 | tee gpurun_out/r5/f2_resunit_tile1.txt
