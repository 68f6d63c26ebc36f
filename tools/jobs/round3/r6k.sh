#!/bin/bash
# round 3, job 6k: q|k|v slice sums consumed by the attention kernels (wide sessions) — parity and frame times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "b32_b64" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_equals_single" 2>&1 | tail -2
for B in 48 64; do
  for v in "" "Q3_WIDE_NO_QKV_FUSE=1"; do
    echo "== B=$B $v"; env $v python tools/prof_run.py 1.7b $B 120 2>&1 | tail -1
  done
done
