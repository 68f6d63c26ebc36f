#!/bin/bash
# r2n: gemm3 schedule variants
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2n.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 300 python tools/prof_prefill.py 1.7b 4096 1 2>&1 | tail -2 >> $O; }
run Q3_GEMM3_V=1
run Q3_GEMM3_V=2
run Q3_GEMM3_V=0
timeout 600 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "prefill_4k" 2>&1 | tail -2 >> $O
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2n_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_prefill.py 1.7b 4096 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r2n_prof 3 2>&1 | grep -i "gemm3\|x3\|kernel " >> $O
cat $O
