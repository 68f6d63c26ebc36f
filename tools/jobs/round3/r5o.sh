#!/bin/bash
# round 3, job o: eight-row workgroups for the N = 1024 split-K projections: full suite + A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
for rep in 1 2; do
  for e in "Q3_SK2_NO_RH=1" "Q3_X=1"; do echo "== $e"; env $e python tools/prof_run.py 1.7b 8 300 | tail -1; done
done
python tools/prof_run.py 0.6b 8 300 | tail -1
