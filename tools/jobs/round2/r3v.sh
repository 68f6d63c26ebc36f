#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3v.txt; : > $O
for n in 100 200 500 1000 2039 4096; do
  echo "== $n ksplit=1: $(env Q3_GEMM3_KSPLIT=1 timeout 300 python tools/prof_prefill.py 1.7b $n 1 2>&1 | tail -1 | cut -c1-100)" >> $O
  echo "== $n auto    : $(timeout 300 python tools/prof_prefill.py 1.7b $n 1 2>&1 | tail -1 | cut -c1-100)" >> $O
done
echo "== 0.6b 500: $(timeout 300 python tools/prof_prefill.py 0.6b 500 1 2>&1 | tail -1 | cut -c1-100)" >> $O
timeout 900 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "prefill_4k" 2>&1 | tail -2 >> $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gemm_prefill or prefill or long_prompt" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 >> $O
cat $O
