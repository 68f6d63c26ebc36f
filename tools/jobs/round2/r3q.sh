#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3q.txt; : > $O
flt() { grep -v "^  File\|Extension modules\|RCCL\|Librccl\|HIP version\|ROCm version\|Hostname\|^$" "$1" | tail -4 | cut -c1-200; }
timeout 900 python -m pytest tests/test_bench_config_parity.py tests/test_c_host.py tests/test_cli.py tests/test_examples.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r3q_a.log 2>&1; echo "A (5 files) rc=$?" >> $O; flt gpurun_out/r3q_a.log >> $O
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3q_b.log 2>&1; echo "B (whole dir) rc=$?" >> $O; flt gpurun_out/r3q_b.log >> $O
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3q_c.log 2>&1; echo "C (whole dir again) rc=$?" >> $O; flt gpurun_out/r3q_c.log >> $O
cat $O
