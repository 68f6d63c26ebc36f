#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3j.txt; : > $O
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "soak or streaming_continuous or error_contract" > gpurun_out/r3j_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 gpurun_out/r3j_$i.log | cut -c1-100)" >> $O
  if [ $rc -ne 0 ]; then grep -v "^  File\|RCCL\|Librccl" gpurun_out/r3j_$i.log | tail -30 >> $O; fi
done
cat $O
