#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3u.txt; : > $O
for n in 200 500 1000; do
  for e in "Q3_X=1" "Q3_GEMM_GEO1=1" "Q3_GEMM_GEO=3" "Q3_PREFILL_GEMM_MIN=0"; do
    echo "== $n $e: $(env $e timeout 300 python tools/prof_prefill.py 1.7b $n 1 2>&1 | tail -1 | cut -c1-90)" >> $O
  done
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3u_prof -o pf -- python $GRAFT_REPO_ROOT/tools/prof_prefill.py 1.7b 500 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/prof_db.py gpurun_out/r3u_prof 3 2>&1 | grep -v "gemv\|gather_rows\|copyBuffer\|attn_fused\|cp_gather" | head -14 >> $O; rm -rf gpurun_out/r3u_prof
cat $O
