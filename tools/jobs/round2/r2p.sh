#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r2p.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 300 python tools/prof_prefill.py 1.7b 4096 1 2>&1 | tail -1 >> $O; }
run Q3_GEMM3_SUP=8,4 Q3_GEMM3_NT=1
run Q3_GEMM3_SUP=8,4
run Q3_GEMM3_SUP=4,8
run Q3_GEMM3_SUP=16,2
run Q3_GEMM3_SUP=8,8
run Q3_GEMM3_SUP=16,4
run Q3_GEMM3_SUP=32,1
run Q3_GEMM3_SUP=8,2
run Q3_GEMM3_SUP=8,6
cat $O
