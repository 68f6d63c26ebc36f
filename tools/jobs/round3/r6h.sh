#!/bin/bash
# round 3, job 6h: two-addend split-K kernel over 16-row blocks for wide sessions — parity at B = 17/33/64 (tiny) and 32/64 (1.7B), frame times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_equals_single or linear_matches" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "b32_b64" 2>&1 | tail -5
for B in 32 48 64; do
  for v in "" "Q3_WIDE_NO_SK2=1"; do
    echo "== B=$B $v"; env $v python tools/prof_run.py 1.7b $B 120 2>&1 | tail -1
  done
done
