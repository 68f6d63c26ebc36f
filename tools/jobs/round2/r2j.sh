#!/bin/bash
# round 2, GPU call J: bench modes for BASELINE's other configurations + full GPU suite
O=gpurun_out/r2j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -n 4 $O/pytest_all.log | cut -c1-200
run() { tag=$1; shift; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-batches "" "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag rc=$?"
  python - $tag <<'PY'
import json, sys
try:
    d=json.loads(open(f"gpurun_out/r2j/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "fps", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), d["stage_ms"], "lat", {k: round(v,2) for k,v in d["latency"].items() if isinstance(v,(int,float))})
except Exception as e:
    print(sys.argv[1], "ERR", e, open(f"gpurun_out/r2j/bench_{sys.argv[1]}.err").read()[-400:])
PY
}
run cfg1_0.6b_b1 --model 0.6b --batch 1
run cfg3_greedy --sampling greedy
run cfg4_voicedesign4k_b1 --workload voicedesign4k --batch 1
run xvector_b8 --workload xvector
