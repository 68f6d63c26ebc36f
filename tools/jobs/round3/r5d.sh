#!/bin/bash
# round 3, job d: split-K-2 projections: parity subset, frame A/B, timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -m gpu -x -q 2>&1 | tail -8
for rep in 1 2; do
  for e in "Q3_NO_KSPLIT=1" "Q3_X=1"; do echo "== $e"; env $e python tools/prof_run.py 1.7b 8 300 | tail -1; done
done
timeout 600 python tools/trace_frame.py 1.7b 8 64 512 --full > gpurun_out/r5d_trace_b8.txt 2>&1
grep -A22 "mean per kernel" gpurun_out/r5d_trace_b8.txt | cut -c1-250
tail -1 gpurun_out/r5d_trace_b8.txt
