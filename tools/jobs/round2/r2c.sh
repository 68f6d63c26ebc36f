#!/bin/bash
# round 2, GPU call C: speech encoder (Mimi) parity, the full GPU suite after the prefill / session / API changes, bench line.
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_speech_encoder.py tests/test_cli.py -m gpu -q > $O/pytest_mimi.log 2>&1; echo "pytest mimi rc=$?" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_speech_encoder.py > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -n 25 $O/pytest_mimi.log; tail -n 5 $O/pytest_all.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2c/bench.json").read().strip().splitlines()[-1])
print("fps", round(d["value"],1), "stage", d["stage_ms"], "roof", round(d["roofline"]["frac"],3), "insitu", round(d["roofline"]["in_situ"]["frac"],3), "lat", d["latency"], "cpu", d["cpu_baseline"])
PY
