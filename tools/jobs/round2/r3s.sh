#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3s.txt; : > $O
env Q3_GEMM_NO_FUSED_PLANES=1 timeout 300 python tools/prof_prefill.py 1.7b 4096 1 2>&1 | tail -1 >> $O
timeout 300 python tools/prof_prefill.py 1.7b 4096 1 2>&1 | tail -2 >> $O
timeout 300 python tools/prof_prefill.py 0.6b 4096 1 2>&1 | tail -1 >> $O
timeout 300 python tools/prof_prefill.py 1.7b 2039 2 2>&1 | tail -1 >> $O
timeout 300 python tools/prof_prefill.py 1.7b 1000 1 2>&1 | tail -1 >> $O
timeout 900 python -m pytest tests/test_bench_config_parity.py -q -x -m gpu -k "prefill_4k" 2>&1 | tail -2 >> $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gemm_prefill or prefill or long_prompt" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 >> $O
cat $O
