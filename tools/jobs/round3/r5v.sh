#!/bin/bash
# round 3, job v: full GPU suite after the vocoder changes + bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3
timeout 900 python bench.py --steps 3 --warmup 1 --no-other-configs 2>&1 | tail -1 > gpurun_out/r5v_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r5v_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k: d.get(k) for k in ('stages', 'stage_ms')})"
