#!/bin/bash
# round 3, job 6t: final per-node timelines (trace build) at B = 8 and B = 64
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6t; export TMPDIR=/tmp
timeout 600 python tools/trace_frame.py 1.7b 8 64 512 --full > gpurun_out/r6t/trace_b8.txt 2>&1; tail -3 gpurun_out/r6t/trace_b8.txt | cut -c1-200
timeout 600 python tools/trace_frame.py 1.7b 64 64 512 > gpurun_out/r6t/trace_b64.txt 2>&1; grep -A22 "mean per kernel" gpurun_out/r6t/trace_b64.txt | cut -c1-200; tail -2 gpurun_out/r6t/trace_b64.txt | cut -c1-200
