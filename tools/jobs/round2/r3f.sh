#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3f.txt; : > $O
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-batches "16,32,64" --ttfa-reps 2 > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err
python - >> $O <<'PY'
import json
d=json.loads(open("gpurun_out/r3f_bench.json").read().strip().splitlines()[-1])
print("fps", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), d["stage_ms"], {k: round(v,2) for k,v in d["latency"].items() if isinstance(v,(int,float))})
print({k: round(v["frames_per_s"],1) for k,v in d["other_batches"].items()})
PY
timeout 3000 python -m pytest tests -q -x -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 >> $O
cat $O
