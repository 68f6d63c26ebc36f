#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear or teacher or free_run or fused" 2>&1 | tail -3 | tee gpurun_out/r5/c2_parity.txt
timeout 900 python tools/dev/env_ab.py "Q3_GEMV_NO_FULL=1" "" --batch 8 --frames 300 2>&1 | tee gpurun_out/r5/c2_ab_b8.txt
