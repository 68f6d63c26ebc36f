#!/bin/bash
# round 6 / C4: per-frame queue lock — two sessions on two host threads, the batcher / streaming tests, frame time unchanged
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests -m gpu -x -q -k "frame_submission or batcher or stream or chunk or continuous or stages_a_4k or free_run" > gpurun_out/r6/c4_tests.txt 2>&1; tail -4 gpurun_out/r6/c4_tests.txt
python tools/dev/aql_ab.py --batch 8 --frames 640 --modes 3,3 2>&1 | grep -v WARNING > gpurun_out/r6/c4_frame.txt; cat gpurun_out/r6/c4_frame.txt
