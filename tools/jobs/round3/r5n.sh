#!/bin/bash
# round 3, job n: 24-row gate/up kernel (k_gemv_gu24): parity + A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_config_parity.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2; do
  for e in "Q3_GEMV_NO_GU24=1" "Q3_X=1"; do echo "== $e"; env $e python tools/prof_run.py 1.7b 8 300 | tail -1; done
done
for e in "Q3_GEMV_NO_GU24=1" "Q3_X=1"; do echo "== $e"; env $e Q3_BENCH_M=1,8,16 python tools/bench_kernels.py 2>&1 | cut -c1-200 | grep "talker gate"; done
