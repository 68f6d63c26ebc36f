#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
Q3_REPLACE_TIMING=1 python tools/dev/time_replace.py 2>&1 | tail -32
