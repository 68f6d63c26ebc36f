#!/bin/bash
# round 6 / C2: the tree after the next-node experiment was removed: frame time back where it was, soak of the native batcher with
# the prefill worker (600 requests of all four prompt kinds, 24 in the system; 24 sampled results against batch-1 runs), with and without
# a page limit.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python tools/dev/aql_ab.py --batch 8 --frames 640 --modes 3,0,3 2>&1 | grep -v WARNING > gpurun_out/r6/c2_frame.txt; cat gpurun_out/r6/c2_frame.txt
timeout 900 python tools/dev/soak_batcher.py 600 30 > gpurun_out/r6/c2_soak.txt 2>&1; tail -4 gpurun_out/r6/c2_soak.txt
timeout 900 python tools/dev/soak_batcher.py 300 200 > gpurun_out/r6/c2_soak_long.txt 2>&1; tail -3 gpurun_out/r6/c2_soak_long.txt
timeout 900 python tools/dev/soak_batcher.py 300 30 12 > gpurun_out/r6/c2_soak_limit.txt 2>&1; tail -3 gpurun_out/r6/c2_soak_limit.txt
