#!/bin/bash
# round 3, job 6y: GEMM path (with q|k|v slice sums consumed by the attention) from 17 rows instead of 33?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for B in 20 24 32; do
  for v in "" "Q3_WIDE_GEMM_MIN=17"; do echo "== B=$B $v"; env $v python tools/prof_run.py 1.7b $B 120 2>&1 | tail -1; done
done
