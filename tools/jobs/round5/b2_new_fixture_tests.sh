#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests/test_bench_config_parity.py -x -q -m gpu -k "frames_behind_4k or config3 or ragged_first_batch_1_7b or 640_frames or b32_b64" 2>&1 | tail -30 | tee gpurun_out/r5/b2_new_fixture_tests.txt
