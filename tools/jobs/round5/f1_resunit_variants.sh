#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
for v in "" occ2 rslate_ring2; do
  if [ -n "$v" ]; then export Q3TTS_LIB=$PWD/build/libq3tts_$v.so; else unset Q3TTS_LIB; fi
  bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
  echo "== ${v:-default}: $(head -1 gpurun_out/vocprof/vocoder_T640.txt) | $(grep k_resunit gpurun_out/vocprof/vocoder_T640.txt | head -1)"
done 2>&1 | tee gpurun_out/r5/f1_resunit_variants.txt
