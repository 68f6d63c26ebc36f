#!/bin/bash
# round 2, GPU call E: K-slice rotation (L2 hot-spot theory): rotated-x stage probe, GEMV sweep and bench with / without rotation
O=gpurun_out/r2e; mkdir -p $O
timeout 300 build/grid_barrier 2>&1 | grep "^D" > $O/stage_probe_rot.txt
Q3_BENCH_M=8 timeout 600 python tools/bench_kernels.py > $O/gemv_rot.txt 2>&1
Q3TTS_LIB=build/libq3tts_norot.so Q3_BENCH_M=8 timeout 600 python tools/bench_kernels.py > $O/gemv_norot.txt 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_rot.json 2> $O/bench_rot.err
Q3TTS_LIB=build/libq3tts_norot.so timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_norot.json 2> $O/bench_norot.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_config_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
cat $O/stage_probe_rot.txt; echo == rot; cat $O/gemv_rot.txt; echo == norot; cat $O/gemv_norot.txt; tail -n 3 $O/pytest.log
python - <<'PY'
import json
for n in ("rot","norot"):
    try:
        d=json.loads(open(f"gpurun_out/r2e/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "fps", round(d["value"],1), "gen ms", round(d["stage_ms"]["generation_ms"],1), "roof", round(d["roofline"]["frac"],3), "b1 ms/frame", round(d["latency"].get("b1_ms_per_frame",0),3), "ttfa", round(d["latency"].get("ttfa_ms_p50",0),2))
    except Exception as e:
        print(n, "ERR", e)
PY
