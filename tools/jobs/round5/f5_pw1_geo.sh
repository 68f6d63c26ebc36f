#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
for g in 0 1 2; do
  Q3_VOC_MIN_NS=100000 Q3_CONV_PW1_GEO=$g bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
  echo "== geo $g: $(head -1 gpurun_out/vocprof/vocoder_T640.txt)"; grep -E "^ *(8[2-8]) " gpurun_out/vocprof/vocoder_T640.txt | cut -c1-110
done 2>&1 | tee gpurun_out/r5/f5_pw1_geo.txt
