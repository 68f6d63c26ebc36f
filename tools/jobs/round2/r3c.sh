#!/bin/bash
# r3c: decode attention with unconditional 3-deep K/V requests: frame time + parity
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3c.txt; : > $O
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-batches "" > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err
python - >> $O <<'PY'
import json
d=json.loads(open("gpurun_out/r3c_bench.json").read().strip().splitlines()[-1])
print("fps", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), d["stage_ms"], {k: round(v,2) for k,v in d["latency"].items() if isinstance(v,(int,float))})
PY
timeout 2400 python -m pytest tests/test_bench_config_parity.py tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -4 >> $O
cat $O
