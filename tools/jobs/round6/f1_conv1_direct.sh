#!/bin/bash
# round 6 / F1: k_conv1_direct vs k_conv_bf16x3 for the decoder blocks' 1x1 residual convs: bit-identity and time at 640 / 128 / 10 frames,
# kernel table of a decode with it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python tools/dev/conv1_direct_check.py 640 5 > gpurun_out/r6/f1_check.txt 2>&1
python tools/dev/conv1_direct_check.py 128 5 >> gpurun_out/r6/f1_check.txt 2>&1
python tools/dev/conv1_direct_check.py 10 5 >> gpurun_out/r6/f1_check.txt 2>&1
cat gpurun_out/r6/f1_check.txt
Q3_CONV1_DIRECT=1 bash tools/prof_vocoder.sh 640 > /dev/null 2>&1; head -22 gpurun_out/vocprof/vocoder_T640.txt | cut -c1-120; cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/r6/f1_vocoder_T640_direct.txt
