#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3l.txt; : > $O
LIBC_FATAL_STDERR_=1 MALLOC_CHECK_=3 timeout 2400 python -X faulthandler -m pytest tests/test_bench_config_parity.py tests/test_c_host.py tests/test_cli.py tests/test_examples.py tests/test_gpu_parity.py -q -m gpu --deselect tests/test_examples.py::test_examples_tts_end_to_end > gpurun_out/r3l_b.log 2>&1; echo "rc=$?" >> $O
grep -v "^  File \"/usr\|RCCL\|Librccl\|Extension modules" gpurun_out/r3l_b.log | tail -25 >> $O
cat $O
