#!/bin/bash
# round 2, GPU call A: full GPU suite (incl. the bench-configuration parity tests), grid-barrier / persistent-stage probe,
# GEMV micro-benchmarks (half-row x loads, producer-side RMSNorm), A/B bench runs.
O=gpurun_out/r2a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_config_parity.py > $O/pytest_base.log 2>&1; echo "pytest base rc=$?" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests/test_bench_config_parity.py -m gpu -q > $O/pytest_benchcfg.log 2>&1; echo "pytest benchcfg rc=$?" | tee -a $O/summary.txt
timeout 300 build/grid_barrier > $O/grid_barrier.txt 2>&1; echo "grid_barrier rc=$?" | tee -a $O/summary.txt
Q3_BENCH_M=8 timeout 600 python tools/bench_kernels.py > $O/gemv_half.txt 2>&1
Q3_BENCH_M=8 Q3_GEMV_NO_HALF=1 timeout 600 python tools/bench_kernels.py > $O/gemv_nohalf.txt 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; echo "bench new rc=$?" | tee -a $O/summary.txt
Q3_NO_PRENORM=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_noprenorm.json 2> $O/bench_noprenorm.err
Q3_NO_PRENORM=1 Q3_GEMV_NO_HALF=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_old.json 2> $O/bench_old.err
tail -3 $O/pytest_base.log $O/pytest_benchcfg.log
cat $O/grid_barrier.txt | tail -40
python - <<'PY'
import json
for n in ("new","noprenorm","old"):
    try:
        d=json.loads(open(f"gpurun_out/r2a/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "fps", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), "stage", d["stage_ms"], "roof", round(d["roofline"]["frac"],3), "b1 ms/frame", d["latency"].get("b1_ms_per_frame"), "ttfa", d["latency"].get("ttfa_ms_p50"))
    except Exception as e:
        print(n, "ERR", e)
PY
