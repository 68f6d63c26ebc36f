#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for e in "Q3_X=1" "Q3_GEMM3_NOB=1"; do
  cd /tmp && env $e rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4f_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_prefill.py 1.7b 4096 1 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT; echo "== $e"; python tools/prof_db.py gpurun_out/r4f_prof 3 2>&1 | grep "gemm3\|x3"; rm -rf gpurun_out/r4f_prof
done
