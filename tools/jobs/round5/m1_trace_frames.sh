#!/bin/bash
# per-node timeline of one captured frame with the round-5 kernels (-DQ3_TRACE build: bash tools/trace_build.sh first): B = 8 and B = 1, frame 300
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 600 python tools/trace_frame.py 1.7b 8 300 512 > gpurun_out/r5/r5_trace_frame_b8.txt 2>&1
timeout 600 python tools/trace_frame.py 1.7b 1 300 512 > gpurun_out/r5/r5_trace_frame_b1.txt 2>&1
head -30 gpurun_out/r5/r5_trace_frame_b8.txt; tail -12 gpurun_out/r5/r5_trace_frame_b8.txt; tail -12 gpurun_out/r5/r5_trace_frame_b1.txt
