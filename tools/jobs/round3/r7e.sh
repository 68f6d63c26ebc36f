#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 900 python bench.py --no-cpu-baseline --also-batches "" > gpurun_out/r7e_bench.json 2> gpurun_out/r7e_bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('gpurun_out/r7e_bench.json').read().strip().splitlines()[-1]); e=d['eos_mix']; print(d['value'], {k: round(e[k],1) for k in e if 'per_s' in k} if 'error' not in e else e)"
