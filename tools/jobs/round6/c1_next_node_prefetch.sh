#!/bin/bash
# round 6 / C1: next-node prefetch — every node of the frame is told where the next node's argument block, kernel descriptor and code
# entry live, and one wave per XCD touches those lines while the node's own operands are in flight. A/B against the same library with
# the fields left zero (Q3_KARG_PREFETCH=0), B = 8 at 640 frames, B = 1 (1.7B, 0.6B); parity tests that replay frames.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python tools/dev/aql_ab.py --batch 8 --frames 640 --modes 3/Q3_KARG_PREFETCH=0,3,3/Q3_KARG_PREFETCH=0,3 2>&1 | grep -v WARNING > gpurun_out/r6/c1_next_node_ab.txt
python tools/dev/aql_ab.py --batch 1 --frames 300 --modes 3/Q3_KARG_PREFETCH=0,3,3/Q3_KARG_PREFETCH=0,3 2>&1 | grep -v WARNING >> gpurun_out/r6/c1_next_node_ab.txt
python tools/dev/aql_ab.py --model 0.6b --batch 1 --frames 300 --modes 3/Q3_KARG_PREFETCH=0,3,3/Q3_KARG_PREFETCH=0,3 2>&1 | grep -v WARNING >> gpurun_out/r6/c1_next_node_ab.txt
cat gpurun_out/r6/c1_next_node_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "free_run or b8_b16 or frame_submission or variants or stream or 640_frames or teacher_forced" > gpurun_out/r6/c1_tests.txt 2>&1
tail -4 gpurun_out/r6/c1_tests.txt
