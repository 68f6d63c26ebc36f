#!/bin/bash
# round 3, job 6f: batcher running in pieces that end at row limits
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batcher or fuzz or replace" 2>&1 | tail -3
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r6f_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r6f_bench.json')); e=d.get('eos_mix'); print(d['value'], {k: e[k] for k in e if 'per_s' in k})"
