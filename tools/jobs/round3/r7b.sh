#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/dev/soak_batcher.py 600 2>&1 | tail -10
