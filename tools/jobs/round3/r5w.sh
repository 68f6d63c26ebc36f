#!/bin/bash
# round 3, job w: vocoder geometry knob Q3_CONV_TM4 = 1 (default) / 2 / 3 per kernel (the 128-co layers and their transposed convs)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for g in 2 3; do
  Q3_CONV_TM4=$g bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
  echo "== Q3_CONV_TM4=$g"; head -2 gpurun_out/vocprof/vocoder_T640.txt | tail -1; grep "launch order" -A60 gpurun_out/vocprof/vocoder_T640.txt | grep " 9[0-9] \|10[0-3] " | cut -c1-130
done
