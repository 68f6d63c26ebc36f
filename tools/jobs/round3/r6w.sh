#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "batcher_1_7b or continuous" 2>&1 | tail -3
