#!/bin/bash
# round 3, job 6o: frame-loop streams cached per model — TTFA, streaming / session tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or replace or batcher or fuzz or chunk" 2>&1 | tail -3
python tools/prof_ttfa.py 2>&1 | tail -6
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --also-batches "" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['latency'])"
