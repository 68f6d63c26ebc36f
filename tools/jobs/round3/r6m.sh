#!/bin/bash
# round 3, job 6m: B = 8 frame time after separating the row-block sk2 instantiation; wide sessions unchanged?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do python tools/prof_run.py 1.7b 8 300 2>&1 | tail -1; done
python tools/prof_run.py 1.7b 64 120 2>&1 | tail -1
python tools/prof_run.py 1.7b 1 300 2>&1 | tail -1
