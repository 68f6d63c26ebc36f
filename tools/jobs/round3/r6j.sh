#!/bin/bash
# round 3, job 6j: wide sk2 without the talker down-proj beyond 32 rows — frame times, parity, B = 64 step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for B in 32 48 64; do echo "== B=$B"; python tools/prof_run.py 1.7b $B 120 2>&1 | tail -1; done
timeout 1500 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "b32_b64" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_equals_single" 2>&1 | tail -2
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 > gpurun_out/r6j_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r6j_bench.json')); print(d['value'], d['other_batches'])"
