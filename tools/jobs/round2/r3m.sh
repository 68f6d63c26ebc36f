#!/bin/bash
# bisect the order-dependent crash / example length failure over library variants (v0 = working tree ... v4 = before the attention rewrite)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3m.txt; : > $O
for v in 4 3 2 1 0; do
  Q3TTS_LIB=$GRAFT_REPO_ROOT/build/variants/libq3_v$v.so LIBC_FATAL_STDERR_=1 timeout 900 python -m pytest tests/test_bench_config_parity.py tests/test_c_host.py tests/test_cli.py tests/test_examples.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r3m_v$v.log 2>&1; rc=$?
  echo "v$v rc=$rc: $(grep -v "RCCL\|Librccl\|HIP version\|ROCm version\|Hostname\|Extension\|^$\|^  File" gpurun_out/r3m_v$v.log | tail -3 | tr '\n' '|' | cut -c1-300)" >> $O
done
cat $O
