#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3e.txt; : > $O
for dp in 2 4 8; do
  env Q3_DECODE_PAIRS=$dp timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-batches "" --ttfa-reps 1 > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err
  python - $dp >> $O <<'PY'
import json, sys
d=json.loads(open("gpurun_out/r3e_bench.json").read().strip().splitlines()[-1])
print("pairs", sys.argv[1], "fps", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), d["stage_ms"])
PY
done
cat $O
