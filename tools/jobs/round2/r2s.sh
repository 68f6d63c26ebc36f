#!/bin/bash
# r2s: vocoder with batched staging loads
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2s.txt; : > $O
timeout 300 python tools/prof_decode.py 640 5 2>&1 | tail -1 >> $O
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2s_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_decode.py 640 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r2s_prof 4 2>&1 | grep -v gemv | head -22 >> $O
timeout 900 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "vocoder" 2>&1 | tail -3 >> $O
cat $O
