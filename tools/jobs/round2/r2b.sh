#!/bin/bash
# round 2, GPU call B: correctness of the software-pipelined GEMV + kernel-choice sweeps (LDS-staged vs register-direct with
# half-row loads, 4-row vs 16-row tiles for N = 1024, 8 vs 16 waves for long K) + A/B bench runs.
O=gpurun_out/r2b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
for v in default nolds big8 nohalf; do
  case $v in default) E="";; nolds) E="Q3_GEMV_NO_LDS=1";; big8) E="Q3_GEMV_BIG8=1";; nohalf) E="Q3_GEMV_NO_HALF=1";; esac
  env $E Q3_BENCH_M=8 timeout 600 python tools/bench_kernels.py > $O/gemv_$v.txt 2>&1
done
Q3TTS_LIB=build/libq3tts_nopipe.so Q3_BENCH_M=8 timeout 600 python tools/bench_kernels.py > $O/gemv_nopipe.txt 2>&1
Q3_BENCH_M=1,16 timeout 600 python tools/bench_kernels.py > $O/gemv_m1_m16.txt 2>&1
run_bench() { env $2 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?" >> $O/summary.txt; }
run_bench new ""
run_bench nomfma4 "Q3_GEMV_NO_MFMA4=1"
run_bench nolds "Q3_GEMV_NO_LDS=1"
run_bench big8 "Q3_GEMV_BIG8=1"
run_bench nopipe "Q3TTS_LIB=build/libq3tts_nopipe.so"
tail -n 3 $O/pytest.log
for v in default nolds big8 nopipe; do echo "== $v"; cat $O/gemv_$v.txt; done
python - <<'PY'
import json
for n in ("new","nomfma4","nolds","big8","nopipe"):
    try:
        d=json.loads(open(f"gpurun_out/r2b/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "fps", round(d["value"],1), "gen ms", round(d["stage_ms"]["generation_ms"],1), "roof", round(d["roofline"]["frac"],3), "insitu", round(d["roofline"]["in_situ"]["frac"],3), "launches", d["roofline"]["launches_per_frame"], "b1 ms/frame", round(d["latency"].get("b1_ms_per_frame",0),3), "ttfa", round(d["latency"].get("ttfa_ms_p50",0),2))
    except Exception as e:
        print(n, "ERR", e)
PY
