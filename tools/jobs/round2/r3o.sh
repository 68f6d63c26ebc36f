#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/res_trace.txt
Q3_RES_TRACE=$GRAFT_REPO_ROOT/gpurun_out/res_trace.txt timeout 900 python -m pytest tests/test_bench_config_parity.py tests/test_c_host.py tests/test_cli.py tests/test_examples.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r3o_a.log 2>&1; echo "rc=$?"
awk 'NR%6==1' gpurun_out/res_trace.txt | cut -c1-160 | tail -30
tail -3 gpurun_out/res_trace.txt | cut -c1-160
cat /proc/sys/vm/max_map_count; ulimit -n
