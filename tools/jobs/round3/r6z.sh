#!/bin/bash
# round 3, job 6z: GEMM path from 17 rows — parity (tiny 17/33/64 rows, 1.7B 32/64 rows, determinism), whole-step numbers
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch_equals_single or linear_matches" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_bench_config_parity.py -m gpu -x -q -k "b32_b64 or deterministic or b8_b16" 2>&1 | tail -3
timeout 1500 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --also-batches 16,20,24,32,48,64 2>&1 | tail -1 > gpurun_out/r6z_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r6z_bench.json')); print(d['value']); [print(k, round(v['frames_per_s'],1), round(v['ms_per_frame'],3)) for k,v in d['other_batches'].items()]"
