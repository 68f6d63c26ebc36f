#!/bin/bash
# round 6 / M1: repeatability of the threaded paths on the last tree: the frame-submission / concurrency tests three times over, the batcher
# tests twice, the overlap path's exactness tests, a soak of 1 200 requests with every finished row vocoded by the decode worker
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for i in 1 2 3; do timeout 900 python -m pytest tests/test_frame_submission.py -m gpu -q 2>&1 | tail -1; done > gpurun_out/r6/m1_stress.txt 2>&1
for i in 1 2; do timeout 1200 python -m pytest tests -m gpu -q -k "batcher or continuous or stages_a_4k or c_host" 2>&1 | tail -1; done >> gpurun_out/r6/m1_stress.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "overlapped or side_by_side" 2>&1 | tail -1 >> gpurun_out/r6/m1_stress.txt
timeout 1500 python tools/dev/soak_batcher.py 1200 30 2>&1 | tail -2 >> gpurun_out/r6/m1_stress.txt
cat gpurun_out/r6/m1_stress.txt
