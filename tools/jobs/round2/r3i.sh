#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -v -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Librccl" > gpurun_out/r3i_full.txt
grep -n "Fatal\|fault\|Segmentation\|Aborted\|FAILED\|passed\|failed" gpurun_out/r3i_full.txt | head -20 > gpurun_out/r3i.txt
grep -n "PASSED" gpurun_out/r3i_full.txt | tail -3 >> gpurun_out/r3i.txt
grep -n -A25 "Fatal Python" gpurun_out/r3i_full.txt | head -60 >> gpurun_out/r3i.txt
cat gpurun_out/r3i.txt
