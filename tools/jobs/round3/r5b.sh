#!/bin/bash
# round 3, job c: per-node timeline of one frame (Q3_TRACE build)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/trace_frame.py 1.7b 8 64 512 --full > gpurun_out/r5c_trace_b8.txt 2>&1
grep -A40 "mean per kernel" gpurun_out/r5c_trace_b8.txt
