#!/bin/bash
# round 6 / J1: the engine against upstream's own talker loop (hf_frame_loop.npz), and the reworked overlap path's exactness tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "upstream_talker or overlapped or side_by_side" > gpurun_out/r6/j1_tests.txt 2>&1; tail -15 gpurun_out/r6/j1_tests.txt
