#!/bin/bash
# round 3, job 6e: full GPU suite + bench line (eos_mix through the native batcher)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r6e_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r6e_bench.json')); print(d['value'], d['ms_per_step'], d.get('eos_mix'))"
