#!/bin/bash
# r2u: vocoder conv kernel with wave-uniform staging / scalar addressing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2u.txt; : > $O
timeout 300 python tools/prof_decode.py 640 5 2>&1 | tail -1 >> $O
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2u_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_decode.py 640 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r2u_prof 4 2>&1 | grep -v gemv | head -18 >> $O
timeout 900 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "vocoder or streaming" 2>&1 | tail -3 >> $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "decode or stream or codec or vocoder or speaker or speech" 2>&1 | tail -3 >> $O
cat $O
