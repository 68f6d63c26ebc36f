#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 40" -ex "info threads" --args python -m pytest tests/test_bench_config_parity.py tests/test_c_host.py tests/test_cli.py tests/test_examples.py tests/test_gpu_parity.py -q -m gpu -p no:faulthandler > gpurun_out/r3p_gdb.log 2>&1
echo "rc=$?"
grep -n "SIGABRT\|Aborted\|signal" gpurun_out/r3p_gdb.log | head -5
grep -n -A45 "received signal" gpurun_out/r3p_gdb.log | cut -c1-220 | head -80
