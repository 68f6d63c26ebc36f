#!/bin/bash
# round 3, job 6s: where does the split decode attention start to pay (ms/frame per 32-frame chunk, context = 10 + frames)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for ns in 1 2 4 8; do Q3_ATTN_SPLITS=$ns python tools/dev/attn_split_crossover.py 8 12 2>&1 | tail -1; done
for ns in 1 4 16; do Q3_ATTN_SPLITS=$ns python tools/dev/attn_split_crossover.py 1 12 2>&1 | tail -1; done
