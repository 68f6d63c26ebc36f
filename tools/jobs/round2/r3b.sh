#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3b.txt; : > $O
for g in 1 2 3; do
  echo "== Q3_CONV_TM4=$g" >> $O
  env Q3_CONV_TM4=$g timeout 300 python tools/prof_decode.py 640 5 2>&1 | tail -1 >> $O
  cd /tmp && env Q3_CONV_TM4=$g rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3b_prof$g -o pf -- python $GRAFT_REPO_ROOT/tools/prof_decode.py 640 2 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r3b_prof$g 3 2>&1 | grep "Li7E\|ILi2E" >> $O; rm -rf gpurun_out/r3b_prof$g
done
env Q3_CONV_TM4=2 timeout 900 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "vocoder" 2>&1 | tail -2 >> $O
cat $O
