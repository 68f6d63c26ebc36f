#!/bin/bash
# round 3, job 6x: whole-step frames/s at 16 / 32 / 48 / 64 rows in one session (bench.py other_batches)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --also-batches 16,32,48,64 2>&1 | tail -1 > gpurun_out/r6x_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r6x_bench.json')); print(d['value']); [print(k, round(v['frames_per_s'],1), round(v['ms_per_frame'],3), v['stage_ms']) for k,v in d['other_batches'].items()]"
