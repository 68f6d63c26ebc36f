#!/bin/bash
# round 3, job 6a: native batcher test; cost of a swap at the bench configuration
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batcher or fuzz or replace" 2>&1 | tail -12
python tools/dev/time_replace.py 2>&1 | tail -8
