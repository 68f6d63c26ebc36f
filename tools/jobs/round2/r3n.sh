#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3n.txt; : > $O
AMD_LOG_LEVEL=1 timeout 900 python -m pytest tests/test_bench_config_parity.py tests/test_c_host.py tests/test_cli.py tests/test_examples.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r3n_a.log 2>&1; echo "A rc=$?" >> $O
grep -v "^  File \"/usr\|Extension modules\|RCCL\|Librccl" gpurun_out/r3n_a.log | tail -25 | cut -c1-300 >> $O
# which earlier files are needed?
timeout 900 python -m pytest tests/test_examples.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r3n_b.log 2>&1; echo "B (examples + gpu_parity) rc=$?" >> $O
timeout 900 python -m pytest tests/test_bench_config_parity.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r3n_c.log 2>&1; echo "C (bench_config + gpu_parity) rc=$?" >> $O
timeout 900 python -m pytest tests/test_cli.py tests/test_c_host.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r3n_d.log 2>&1; echo "D (cli + c_host + gpu_parity) rc=$?" >> $O
dmesg 2>/dev/null | tail -5 >> $O
cat $O
