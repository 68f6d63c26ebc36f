#!/bin/bash
# r2r: vocoder conv geometry with 128-column wave tiles (T_M = 4): timing + kernel table + parity
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2r.txt; : > $O
for v in 0 1; do
  echo "== Q3_CONV_TM4=$v" >> $O
  env Q3_CONV_TM4=$v timeout 300 python tools/prof_decode.py 640 5 2>&1 | tail -1 >> $O
  cd /tmp && env Q3_CONV_TM4=$v rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2r_prof$v -o pf -- python $GRAFT_REPO_ROOT/tools/prof_decode.py 640 3 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r2r_prof$v 4 2>&1 | head -22 >> $O
done
env Q3_CONV_TM4=1 timeout 900 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "vocoder" 2>&1 | tail -3 >> $O
cat $O
