#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "long_prompt_batch" 2>&1 | tail -40 > gpurun_out/r2w.txt
cat gpurun_out/r2w.txt
