#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2000 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -60 > gpurun_out/r3g.txt
cat gpurun_out/r3g.txt
