#!/bin/bash
# round 6 / H1: the vocoder beside the frame loop with the chip SPLIT — the frames' own queue confined to 256 - n CUs, the decode
# stream to the other n (8 / n of every XCD) — against the plain step (all frames, then the vocoder). Exactness tests first.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "overlapped or side_by_side" > gpurun_out/r6/h1_tests.txt 2>&1; tail -3 gpurun_out/r6/h1_tests.txt
for v in "Q3_DECODE_OVERLAP=0" "Q3_DECODE_OVERLAP=1 Q3_DECODE_CUS=64" "Q3_DECODE_OVERLAP=1 Q3_DECODE_CUS=32" "Q3_DECODE_OVERLAP=1 Q3_DECODE_CUS=96" "Q3_DECODE_OVERLAP=1 Q3_DECODE_CUS=64 Q3_DECODE_SEG=64" "Q3_DECODE_OVERLAP=1 Q3_DECODE_CUS=64 Q3_DECODE_TAIL=16" "Q3_DECODE_OVERLAP=0"; do
  env $v python bench.py --headline-only --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), d['stage_ms'])"
done > gpurun_out/r6/h1_split_overlap.txt 2>&1
cat gpurun_out/r6/h1_split_overlap.txt
