#!/bin/bash
# round 6 / K1: the split-chip overlap (H1) on the WIDE session: at 64 rows the frame's kernels are 64-192 workgroups of 256-512 threads,
# not one per CU of the whole chip, and the vocoder is 23 % of the step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for v in "Q3_DECODE_OVERLAP=0" "Q3_DECODE_OVERLAP=1" "Q3_DECODE_OVERLAP=1 Q3_DECODE_CUS=64" "Q3_DECODE_OVERLAP=1 Q3_DECODE_CUS=96" "Q3_DECODE_OVERLAP=1 Q3_DECODE_CUS=128" "Q3_FRAME_CUS=192" "Q3_FRAME_CUS=160"; do
  env $v python bench.py --headline-only --batch 64 --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), d['stage_ms'])"
done > gpurun_out/r6/k1_split_overlap_b64.txt 2>&1
cat gpurun_out/r6/k1_split_overlap_b64.txt
