#!/bin/bash
# round 3, job 6n: key splits of the decode attention in a 64-row session over the benchmark's 640 frames
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for ns in 1 2 4; do echo "== Q3_ATTN_SPLITS=$ns"; Q3_ATTN_SPLITS=$ns python tools/prof_run.py 1.7b 64 640 2>&1 | tail -1; done
for ns in 1 2 4; do echo "== B=32 Q3_ATTN_SPLITS=$ns"; Q3_ATTN_SPLITS=$ns python tools/prof_run.py 1.7b 32 640 2>&1 | tail -1; done
