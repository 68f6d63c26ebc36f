#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3x.txt; : > $O
for n in 24 40 56 72 88 120; do
  echo "== $n gemm : $(env Q3_PREFILL_GEMM_MIN=1 timeout 300 python tools/prof_prefill.py 1.7b $n 1 2>&1 | tail -1 | cut -c1-80)" >> $O
  echo "== $n steps: $(env Q3_PREFILL_GEMM_MIN=0 timeout 300 python tools/prof_prefill.py 1.7b $n 1 2>&1 | tail -1 | cut -c1-80)" >> $O
done
for n in 40 72; do
  echo "== B=4 $n gemm : $(env Q3_PREFILL_GEMM_MIN=1 timeout 300 python tools/prof_prefill.py 1.7b $n 4 2>&1 | tail -1 | cut -c1-80)" >> $O
  echo "== B=4 $n steps: $(env Q3_PREFILL_GEMM_MIN=0 timeout 300 python tools/prof_prefill.py 1.7b $n 4 2>&1 | tail -1 | cut -c1-80)" >> $O
done
cat $O
