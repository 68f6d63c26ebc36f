#!/bin/bash
# round 2, GPU call F: wide batches (B up to 64): op parity, batch == single, 1.7B B = 32, bench at 8 / 16 / 32 / 64
O=gpurun_out/r2f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "linear or batch_equals" > $O/pytest_wide.log 2>&1; echo "pytest wide rc=$?"
timeout 900 python -m pytest tests/test_bench_config_parity.py -m gpu -q -x -k "b32 or b8_b16" > $O/pytest_b32.log 2>&1; echo "pytest b32 rc=$?"
Q3_BENCH_M=16,32,64 timeout 900 python tools/bench_kernels.py > $O/gemv_wide.txt 2>&1
timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-batches 16,32,64 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -n 15 $O/pytest_wide.log | cut -c1-200; tail -n 15 $O/pytest_b32.log | cut -c1-200; cat $O/gemv_wide.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2f/bench.json").read().strip().splitlines()[-1])
print("B=8 fps", round(d["value"],1), d["stage_ms"]); print(json.dumps(d["other_batches"], indent=1))
PY
