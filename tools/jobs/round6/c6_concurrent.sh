#!/bin/bash
# round 6 / C6: concurrent synthesize calls from four host threads on one model (tests/test_frame_submission.py), three times over
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for i in 1 2 3; do timeout 600 python -m pytest tests/test_frame_submission.py -m gpu -x -q > gpurun_out/r6/c6_tests_$i.txt 2>&1; tail -2 gpurun_out/r6/c6_tests_$i.txt; grep -n "thread [0-9]:" gpurun_out/r6/c6_tests_$i.txt | head -3 | cut -c1-300; done
