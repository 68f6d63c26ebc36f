#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r7d_pytest.txt 2>&1; echo "pytest rc=$?"; grep "passed\|failed" gpurun_out/r7d_pytest.txt | tail -2
