#!/bin/bash
# round 3, job g: full GPU suite with the new parity tests (long context, 640 frames, clone flavours, 2-rank bench), frame times, timeline incl. shader clock
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for rep in 1 2; do python tools/prof_run.py 1.7b 8 300 | tail -1; done
python tools/prof_run.py 1.7b 1 300 | tail -1
timeout 600 python tools/trace_frame.py 1.7b 8 64 512 --full > gpurun_out/r5g_trace_b8.txt 2>&1
grep -A14 "mean per kernel" gpurun_out/r5g_trace_b8.txt | cut -c1-250
grep "fold" gpurun_out/r5g_trace_b8.txt | tail -3 | cut -c1-200
tail -3 gpurun_out/r5g_trace_b8.txt
