#!/bin/bash
# round 3, job s: kernel tables with names — B = 64 frame, one 640-frame vocoder decode in launch order
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/prof_frame.sh 1.7b 64 60 512 2>&1 | tail -40 | cut -c1-170
cp gpurun_out/frameprof/frame_1.7b_b64.txt gpurun_out/r5s_frame_b64.txt 2>/dev/null
bash tools/prof_vocoder.sh 640 > /dev/null 2>&1
cp gpurun_out/vocprof/vocoder_T640.txt gpurun_out/r5s_vocoder_T640.txt; cat gpurun_out/r5s_vocoder_T640.txt | cut -c1-150
