#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3k.txt; : > $O
timeout 1500 python -m pytest tests/test_bench_config_parity.py tests/test_c_host.py tests/test_cli.py tests/test_examples.py tests/test_gpu_parity.py -q -x -m gpu > gpurun_out/r3k_a.log 2>&1; echo "A rc=$?" >> $O
grep -v "^  File\|RCCL\|Librccl\|Extension modules" gpurun_out/r3k_a.log | tail -12 >> $O
AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 timeout 2400 python -m pytest tests/test_bench_config_parity.py tests/test_c_host.py tests/test_cli.py tests/test_examples.py tests/test_gpu_parity.py -q -x -m gpu > gpurun_out/r3k_b.log 2>&1; echo "B(serialized) rc=$?" >> $O
grep -v "^  File\|RCCL\|Librccl\|Extension modules" gpurun_out/r3k_b.log | tail -12 >> $O
cat $O
