#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/prof_run.py 1.7b 8 300 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "free_run or batch_equals or teacher" 2>&1 | tail -2
