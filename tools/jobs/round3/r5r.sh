#!/bin/bash
# round 3, job r: per-row sampler settings — GPU suite + frame time check
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sampling_options or replace or sample or rows_end" 2>&1 | tail -15
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-other-configs 2>&1 | tail -1 > gpurun_out/r5r_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r5r_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('eos_mix'))"
