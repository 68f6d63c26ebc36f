#!/bin/bash
# round 3, job a: GPU suite with the rescaled synthetic decoder + code-predictor attention kernel; frame A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for rep in 1 2; do
  for e in "Q3_NO_CP_ATTN=1" "Q3_X=1"; do echo "== $e"; env $e python tools/prof_run.py 1.7b 8 300 | tail -1; done
done
bash tools/prof_frame.sh 1.7b 8 200 512 2>&1 | tail -60
