#!/bin/bash
# round 6 / B6: why bench.py's eos_mix leg reads 2-3 % below tools/dev/eos_mix_ab.py: the same tool with torch loaded first (its bundled
# HIP runtime then serves libq3tts.so); where a swap's milliseconds go in both (Q3_REPLACE_TIMING); B = 64 vocoder phase at 2 / 4 at a time.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
python tools/dev/eos_mix_ab.py 2 > gpurun_out/r6/b6_eosmix.txt 2>&1
EOS_MIX_TORCH=1 python tools/dev/eos_mix_ab.py 2 2>&1 | sed 's/^/torch first: /' >> gpurun_out/r6/b6_eosmix.txt
Q3_BAT_NO_STAGE=1 Q3_REPLACE_TIMING=1 python tools/dev/eos_mix_ab.py 1 2>&1 | grep "q3 replace" | awk '{k=$3; v[k]+=$4; n[k]++} END {for (k in v) printf "no torch   replace %-8s cumulative mean %.3f ms over %d\n", k, v[k]/n[k], n[k]}' >> gpurun_out/r6/b6_eosmix.txt
EOS_MIX_TORCH=1 Q3_BAT_NO_STAGE=1 Q3_REPLACE_TIMING=1 python tools/dev/eos_mix_ab.py 1 2>&1 | grep "q3 replace" | awk '{k=$3; v[k]+=$4; n[k]++} END {for (k in v) printf "torch first replace %-8s cumulative mean %.3f ms over %d\n", k, v[k]/n[k], n[k]}' >> gpurun_out/r6/b6_eosmix.txt
cat gpurun_out/r6/b6_eosmix.txt
for v in "Q3_DECODE_PAIRS=4" "Q3_DECODE_PAIRS=2"; do
  env $v python bench.py --headline-only --batch 64 --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=64 $v', round(d['value'],1), d['stage_ms'])"
done > gpurun_out/r6/b6_decode_pairs_b64.txt 2>&1
cat gpurun_out/r6/b6_decode_pairs_b64.txt
