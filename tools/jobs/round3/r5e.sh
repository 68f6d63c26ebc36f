#!/bin/bash
# round 3, job e: kernarg preload + zero jobs behind the loads + mfma4 transposing reduction: parity, frame A/B (B = 8, 1), timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for rep in 1 2; do python tools/prof_run.py 1.7b 8 300 | tail -1; done
python tools/prof_run.py 1.7b 1 300 | tail -1
python tools/prof_run.py 0.6b 1 300 | tail -1
timeout 600 python tools/trace_frame.py 1.7b 8 64 512 --full > gpurun_out/r5e_trace_b8.txt 2>&1
grep -A22 "mean per kernel" gpurun_out/r5e_trace_b8.txt | cut -c1-250
tail -1 gpurun_out/r5e_trace_b8.txt
timeout 600 python tools/trace_frame.py 1.7b 1 64 512 --full > gpurun_out/r5e_trace_b1.txt 2>&1
grep -A22 "mean per kernel" gpurun_out/r5e_trace_b1.txt | cut -c1-250
tail -1 gpurun_out/r5e_trace_b1.txt
