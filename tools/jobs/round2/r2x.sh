#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python tools/diag_prefill_det.py 2102 > gpurun_out/r2x.txt 2>&1
cat gpurun_out/r2x.txt
