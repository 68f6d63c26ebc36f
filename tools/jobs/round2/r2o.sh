#!/bin/bash
# r2o: gemm3 super-tile shapes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2o.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 300 python tools/prof_prefill.py 1.7b 4096 1 2>&1 | tail -1 >> $O; }
run Q3_X=1
run Q3_GEMM3_SUP=1,8
run Q3_GEMM3_SUP=2,16
run Q3_GEMM3_SUP=8,4
run Q3_GEMM3_SUP=4,4
run Q3_GEMM3_SUP=2,8
run Q3_GEMM3_V=1
echo "== 2039 x1 default / GEO=3 / GEO=2; 1000 default / GEO=3; 0.6b" >> $O
timeout 300 python tools/prof_prefill.py 1.7b 2039 1 2>&1 | tail -1 >> $O
env Q3_GEMM_GEO=3 timeout 300 python tools/prof_prefill.py 1.7b 2039 1 2>&1 | tail -1 >> $O
env Q3_GEMM_GEO=2 timeout 300 python tools/prof_prefill.py 1.7b 2039 1 2>&1 | tail -1 >> $O
timeout 300 python tools/prof_prefill.py 1.7b 1000 1 2>&1 | tail -1 >> $O
env Q3_GEMM_GEO=3 timeout 300 python tools/prof_prefill.py 1.7b 1000 1 2>&1 | tail -1 >> $O
env Q3_GEMM_GEO=3 timeout 300 python tools/prof_prefill.py 1.7b 500 1 2>&1 | tail -1 >> $O
env Q3_GEMM_GEO=2 timeout 300 python tools/prof_prefill.py 1.7b 500 1 2>&1 | tail -1 >> $O
timeout 300 python tools/prof_prefill.py 0.6b 4096 1 2>&1 | tail -1 >> $O
timeout 600 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "prefill_4k" 2>&1 | tail -2 >> $O
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2o_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_prefill.py 1.7b 4096 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r2o_prof 3 2>&1 | grep -i "gemm3\|x3\|kernel " >> $O
cat $O
