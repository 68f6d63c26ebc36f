#!/bin/bash
# round 3, job i: wide-session GEMM (q3_kernels_wide.hip): parity + B = 32 / 64 frame times against k_gemv_wide
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear_matches_oracle or batch_equals_single or fused_residual" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "b32" 2>&1 | tail -5
for B in 64 32; do
  for e in "Q3_WIDE_NO_GEMM=1" "Q3_X=1"; do echo "== B=$B $e"; env $e python tools/prof_run.py 1.7b $B 120 | tail -1; done
done
Q3_BENCH_M=64 python tools/bench_kernels.py 2>&1 | tail -12; Q3_WIDE_NO_GEMM=1 Q3_BENCH_M=64 python tools/bench_kernels.py 2>&1 | tail -12
