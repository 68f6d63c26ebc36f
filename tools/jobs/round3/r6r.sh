#!/bin/bash
# round 3, job 6r: batcher opening on idle rows — tests, C host, eos_mix
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_c_host.py -m gpu -x -q -k "batcher or c_host" 2>&1 | tail -5
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --also-batches "" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['eos_mix']; print(d['value'], {k: e[k] for k in e if 'per_s' in k or k=='error'})"
