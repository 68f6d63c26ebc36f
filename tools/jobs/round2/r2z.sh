#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2z.txt; : > $O
for sg in 0 1 2 3 4 6; do
  echo "== Q3_CONV_STAGGER=$sg" >> $O
  env Q3_CONV_STAGGER=$sg timeout 300 python tools/prof_decode.py 640 5 2>&1 | tail -1 >> $O
done
cd /tmp && env Q3_CONV_STAGGER=2 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2z_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_decode.py 640 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r2z_prof 4 2>&1 | grep "Li7E\|ILi2E" >> $O
cat $O
