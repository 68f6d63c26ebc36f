#!/bin/bash
# round 3, job z: mixed first batch / 1.7B continuous parity; cost of a swap at the bench configuration
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz or replace" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "continuous" 2>&1 | tail -5
python tools/dev/time_replace.py 2>&1 | tail -8
