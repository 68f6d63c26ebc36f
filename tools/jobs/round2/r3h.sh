#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_parity.py -v -x -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Librccl" | grep "PASSED\|FAILED\|Fatal\|fault\|Error\|error\|test_" | tail -30 > gpurun_out/r3h.txt
cat gpurun_out/r3h.txt
