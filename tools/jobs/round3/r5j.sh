#!/bin/bash
# round 3, job j: wide-session GEMM, second cut (chunk prefetch, 64-row groups, slices of >= 2 chunks)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear_matches_oracle or batch_equals_single" 2>&1 | tail -4
for e in "Q3_WIDE_NO_GEMM=1" "Q3_X=1"; do echo "== B=64 $e"; env $e python tools/prof_run.py 1.7b 64 120 | tail -1; done
echo "== B=48"; python tools/prof_run.py 1.7b 48 120 | tail -1; Q3_WIDE_NO_GEMM=1 python tools/prof_run.py 1.7b 48 120 | tail -1
Q3_BENCH_M=64 python tools/bench_kernels.py 2>&1 | tail -12
