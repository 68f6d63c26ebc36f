#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/dev/soak_batcher.py 600 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
python tools/prof_run.py 1.7b 8 300 | tail -1
