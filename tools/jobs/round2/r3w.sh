#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3w.txt; : > $O
for n in 4096 4096 2039 500; do
  echo "== $n: $(timeout 300 python tools/prof_prefill.py 1.7b $n 1 2>&1 | tail -1 | cut -c1-100)" >> $O
done
cat $O
