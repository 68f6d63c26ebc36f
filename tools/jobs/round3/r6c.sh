#!/bin/bash
# round 3, job 6c: lazy zero-fill at session creation — swap cost, TTFA, continuous tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
Q3_REPLACE_TIMING=1 python tools/dev/time_replace.py 2>&1 | tail -18
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batcher or fuzz or replace or batch_equals or stream" 2>&1 | tail -3
timeout 900 python bench.py --steps 2 --warmup 1 --no-other-configs 2>&1 | tail -1 > gpurun_out/r6c_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r6c_bench.json')); print(d['value'], d['ms_per_step'], d.get('latency'), d.get('eos_mix'))"
