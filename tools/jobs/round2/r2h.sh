#!/bin/bash
# round 2, GPU call H: 16-wave long-K kernel variants (hoisted order / x-first) on talker down; bench A/B
O=gpurun_out/r2h; mkdir -p $O
for v in default HOIST16 XFIRST16; do
  if [ $v = default ]; then L=""; else L="Q3TTS_LIB=build/libq3tts_$v.so"; fi
  env $L Q3_BENCH_M=8 timeout 600 python tools/bench_kernels.py 2>&1 | grep -E "down|talker o|codec head" > $O/gemv_$v.txt
  env $L timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-batches "" > $O/bench_$v.json 2> $O/bench_$v.err
done
for v in default HOIST16 XFIRST16; do echo "== $v"; cat $O/gemv_$v.txt; python - $v <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r2h/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("fps", round(d["value"],1), "gen ms", round(d["stage_ms"]["generation_ms"],1))
PY
done
