#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3a.txt; : > $O
for d in 0 1 2 4 8 3 5 6 7 15; do
  cd /tmp && env Q3_CONV_DBG=$d rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3a_prof$d -o pf -- python $GRAFT_REPO_ROOT/tools/prof_decode.py 640 2 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT; echo "== dbg $d" >> $O; python tools/prof_db.py gpurun_out/r3a_prof$d 3 2>&1 | grep "Li7E" >> $O; rm -rf gpurun_out/r3a_prof$d
done
cat $O
