#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3r_prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --also-batches "" --ttfa-reps 0 --frames 640 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db_grid.py gpurun_out/r3r_prof attn > gpurun_out/r3r.txt 2>&1; python tools/prof_db_grid.py gpurun_out/r3r_prof gemv | head -24 >> gpurun_out/r3r.txt; rm -rf gpurun_out/r3r_prof
cat gpurun_out/r3r.txt
