#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/prof_frame.sh 1.7b 64 60 512 2>&1 | tail -32 | cut -c1-170
cp gpurun_out/frameprof/frame_1.7b_b64.txt gpurun_out/r6i_frame_b64.txt 2>/dev/null
