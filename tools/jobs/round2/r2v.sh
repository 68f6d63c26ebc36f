#!/bin/bash
# r2v: full GPU suite after the prefill / vocoder kernel work
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 > gpurun_out/r2v.txt
cat gpurun_out/r2v.txt
