#!/bin/bash
# round 3, job k: continuous batching tests + whole suite + wide plan re-check
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "own_limit or replace or several_rows" 2>&1 | tail -15
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/prof_run.py 1.7b 64 120 | tail -1
python tools/prof_run.py 1.7b 8 300 | tail -1
Q3_BENCH_M=64 python tools/bench_kernels.py 2>&1 | grep "gate/up\|talker qkv\|talker o"
