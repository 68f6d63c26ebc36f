#!/bin/bash
# round 6 / H2: why the frame loop slows beside a confined vocoder (H1): (a) the frames alone on 224 / 192 / 128 CUs, nothing beside
# them; (b) shader clock and power while frames and vocoder overlap.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for v in "Q3_FRAME_CUS=0" "Q3_FRAME_CUS=224" "Q3_FRAME_CUS=192" "Q3_FRAME_CUS=128"; do
  env $v python bench.py --headline-only --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), d['stage_ms'])"
done > gpurun_out/r6/h2_frames_on_fewer_cus.txt 2>&1
cat gpurun_out/r6/h2_frames_on_fewer_cus.txt
for v in "Q3_DECODE_OVERLAP=0" "Q3_DECODE_OVERLAP=1"; do
  env $v python bench.py --headline-only --steps 6 --warmup 1 > /dev/null 2>&1 &
  pid=$!
  sleep 25
  echo "$v:"
  while kill -0 $pid 2>/dev/null; do rocm-smi --showclocks --showpower | grep -E "sclk|Power \(W\)" | tr '\n' ' '; echo; sleep 0.3; done
done > gpurun_out/r6/h2_clock_overlap.txt 2>&1
awk '/OVERLAP/ {print} /sclk/ {n++; if (n % 4 == 0) print}' gpurun_out/r6/h2_clock_overlap.txt | cut -c1-200 | head -60
