#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "deterministic or b32_b64" 2>&1 | tail -3
