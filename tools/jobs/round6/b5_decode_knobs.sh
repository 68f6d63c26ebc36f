#!/bin/bash
# round 6 / B5: what the vocoder phase of the timed step does with the frame loop on its own queue: utterances decoded 1 / 2 / 4 / 8
# at a time (Q3_DECODE_PAIRS) and 128-frame segments decoded beside the frame loop (Q3_DECODE_OVERLAP=1, lost in round 1) — headline
# step only; then the default bench line with the eos_mix leg moved behind the timed steps.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for v in "Q3_DECODE_PAIRS=4" "Q3_DECODE_PAIRS=8" "Q3_DECODE_PAIRS=2" "Q3_DECODE_OVERLAP=1" "Q3_DECODE_PAIRS=4"; do
  env $v python bench.py --headline-only --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), d['stage_ms'])"
done > gpurun_out/r6/b5_decode_knobs.txt 2>&1
cat gpurun_out/r6/b5_decode_knobs.txt
timeout 600 python -m pytest tests/test_frame_submission.py tests/test_abi.py -m gpu -q > gpurun_out/r6/b5_tests.txt 2>&1; tail -15 gpurun_out/r6/b5_tests.txt
timeout 900 python bench.py > gpurun_out/r6/b5_bench.json 2> gpurun_out/r6/b5_bench.err
python - <<'PY'
import json
p=json.load(open("gpurun_out/r6/b5_bench.json"))
print(p["value"], p["stage_ms"], p["roofline"]["frame_ms"], p["latency"]["ttfa_ms_p50"])
print({k:(round(v) if isinstance(v,float) else v) for k,v in p["eos_mix"].items() if k!="what"})
PY
